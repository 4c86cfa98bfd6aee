#!/bin/bash
# Run argument-validation / kernel-parity / API tests against the UBSan build of libvfx_hip (host side, no recovery:
# a report aborts python).   make -C voicefixer_amd/csrc ubsan   cross-compiles without a GPU; the .so travels with gpurun.
[ -f voicefixer_amd/libvfx_hip_ubsan.so ] || make -C voicefixer_amd/csrc ubsan || exit 1
export VFX_DEV=1 VFX_LIB=$PWD/voicefixer_amd/libvfx_hip_ubsan.so
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
python -m pytest tests/test_ops_gpu.py tests/test_api_gpu.py -q -m gpu -x \
    -k "bad_arguments or test_conv1d or convtr or resblock or conv2d or stft_mel or ragged or restore_inmem_matches_golden or hf_cut or gru" "$@"
