export TMPDIR=/tmp
for f in 0.25 0.5 0.7 1.0; do for m in 1024; do echo -n "N=100 fill $f min $m: "; PASTIX_AMD_QUAD_FILL=$f PASTIX_AMD_QUAD_MIN=$m python tools/dev_bench.py -n 100 --reps 4 2>&1 | tail -1 | cut -c1-75; done; done
for f in 0.5 1.0; do echo -n "N=160 fill $f: "; PASTIX_AMD_QUAD_FILL=$f python tools/dev_bench.py -n 160 --reps 3 2>&1 | tail -1 | cut -c1-75; done
