export TMPDIR=/tmp
for ov in 1 0; do
rm -rf /tmp/le
PASTIX_AMD_OVERLAP=$ov rocprofv3 --kernel-trace --output-format csv -d /tmp/le -- python3 tools/dev_bench.py -n 100 --reps 2 > /tmp/le.log 2>&1
echo "OVERLAP=$ov: $(tail -1 /tmp/le.log | cut -c1-60)"
python3 tools/gap_stats.py /tmp/le
done
