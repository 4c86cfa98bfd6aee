python -m pytest tests/test_model_gpu.py -x -q -m gpu -k ragged 2>&1 | tail -3
