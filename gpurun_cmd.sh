python -m pytest tests -x -q -m gpu -k "hf_cut or mode1" 2>&1 | tail -8
