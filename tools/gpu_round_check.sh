export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -2 | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
