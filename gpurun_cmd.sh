nproc; lscpu | grep "Model name" | head -1
time python bench.py 2>&1 | tail -1
