export TMPDIR=/tmp
for ov in 1 0; do
rm -rf /tmp/zp; PASTIX_AMD_OVERLAP=$ov rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/zp -- python3 bench.py --grid 48 --workload elasticity --steps 3 --warmup 1 --no-cpu-baseline > /tmp/zp.json 2>/dev/null
echo "OVERLAP=$ov $(python3 -c "import json; d=json.loads(open('/tmp/zp.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
head -6 $(find /tmp/zp -name "*kernel_stats.csv" | head -1) | cut -c1-50,90-200
done
