export TMPDIR=/tmp
for leaf in 4 6 8 10 12 16; do for am in 5 12; do
echo -n "leaf $leaf amalg $am: "; python tools/dev_bench.py -n 100 --leaf $leaf --amalg $am --reps 3 2>&1 | awk '/^N=/{printf "%s %s %s | ", $2, $4, $5} /^plan/{printf "%s | ", $3} /^rep 2/{print $6, $7, $9, $10}'
done; done
