#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/t_all.log 2>&1; tail -8 $OUT/t_all.log
