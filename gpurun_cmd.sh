python -m pytest tests -x -q -m gpu 2>&1 | tail -5
for B in 1 32; do
python bench.py --steps 3 --warmup 1 --batch $B --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('B=%d  %.1f xRT  %.1f ms/step  path %.1f TF  dom %s %.1f TF  conv-share %.2f' % (d['config']['batch_per_gpu'], d['value'], d['ms_per_step'], d['path_tflops'], d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['all_conv_kernels']['time_share_of_step']))"
done
