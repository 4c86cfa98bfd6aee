for W in 0 1 0 1; do
  export VFX_WAVES8=$W
  echo "=== WAVES8=$W"
  python tools/conv_bench.py --batch 16 res4_d1 res3_d1 res2_d1 res2_d243 res1_d1 up2 unet2 unet3 2>&1 | grep -v amdgpu.ids
done
VFX_WAVES8=1 python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -1
