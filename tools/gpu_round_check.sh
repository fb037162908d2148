mkdir -p gpurun_out/r02m
for mp in 8 16 32; do for c in 1024 2048 4096; do
echo "maxpieces=$mp chunk=$c"
for n in 100; do
PASTIX_AMD_MAXPIECES=$mp python tools/dev_bench.py -n $n --reps 4 --look $c 2>&1 | tail -1
done
done; done > gpurun_out/r02m/chunks2.txt 2>&1
cat gpurun_out/r02m/chunks2.txt | cut -c1-110
