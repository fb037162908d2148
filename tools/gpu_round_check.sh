for bs in 128 96 64; do for g in 40 48; do
python bench.py --workload elasticity --grid $g --blocksize $bs --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('bs=$bs grid=$g', d['value'], d['ms_per_step'], d['config']['residual'], d['config']['fact_flops'])"
done; done
