python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py --steps 2 --warmup 1 --batch 32 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
