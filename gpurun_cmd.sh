python -m pytest tests -x -q -m gpu 2>&1 | tail -1
for B in 1 32; do
python bench.py --steps 3 --warmup 1 --batch $B --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('B=%d  %.1f xRT  %.1f ms/step  dom %.1f TF  all conv %.1f TF' % (d['config']['batch_per_gpu'], d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['all_conv_kernels']['achieved']))"
done
