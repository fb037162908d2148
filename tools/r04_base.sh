cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for n in 60 80 100 130; do python tools/dev_bench.py -n $n --reps 3 2>&1 | tail -3; done > gpurun_out/r04/base_dev.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04/trace60 -- python3 $GRAFT_REPO_ROOT/tools/dev_bench.py -n 60 --reps 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/r04/trace60/*/*kernel_trace.csv | head -1); python tools/level_timeline.py $f > gpurun_out/r04/level_timeline_d60_base.txt 2>&1
rm -rf gpurun_out/r04/trace60
cat gpurun_out/r04/base_dev.txt gpurun_out/r04/level_timeline_d60_base.txt
