export TMPDIR=/tmp
for n in 100 200; do
rm -rf /tmp/le
rocprofv3 --kernel-trace --output-format csv -d /tmp/le -- python3 tools/dev_bench.py -n $n --reps 2 > /tmp/le.log 2>&1
python3 tools/level_timeline.py /tmp/le 2 > gpurun_out/level_timeline_${n}cube.txt 2>&1
python3 tools/gap_stats.py /tmp/le > gpurun_out/gap_stats_${n}cube.txt 2>&1
python3 tools/leaf_timeline.py /tmp/le 70 > gpurun_out/leaf_timeline_${n}cube.txt 2>&1
done
cat gpurun_out/level_timeline_100cube.txt gpurun_out/gap_stats_200cube.txt
