#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/ -x -q -m gpu -k "single" 2>&1 | tail -3
python bench.py --dtype f32 --grid 100 --no-cpu-baseline --no-other-configs --steps 5 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['residual'], d['roofline']['frac'])"
python bench.py --dtype f32 --no-cpu-baseline --no-other-configs --steps 2 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['residual'], d['roofline']['frac'])"
python bench.py --dtype f32 --grid 100 --facto ldlt --no-cpu-baseline --no-other-configs --steps 3 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['residual'])"
python bench.py --dtype f32 --grid 100 --facto lu --no-cpu-baseline --no-other-configs --steps 3 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['residual'])"
