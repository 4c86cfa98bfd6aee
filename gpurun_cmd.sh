python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -2
python tools/conv_bench.py --batch 8 res4_d1 res4_d27 res4_d2187 res3_d1 res2_d1 res2_d243 res1_d1 res1_d729 pre_k7 cond up1 up2 up3 up4 unet1 unet2 unet3 unet4 unet5 gru_proj 2>&1 | grep -v amdgpu.ids
