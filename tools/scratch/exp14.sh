#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python tools/scratch/lu_diff.py 2>&1 | tail -60
