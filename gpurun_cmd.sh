python -m pytest tests/test_api_gpu.py -x -q -m gpu -k "batch" 2>&1 | tail -2
