#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "categories or interleaved" > $OUT/t_cat_par.log 2>&1; tail -3 $OUT/t_cat_par.log
timeout 1500 python -m pytest tests/test_hyphy_integration.py -x -q -m gpu > $OUT/t_integration.log 2>&1; tail -8 $OUT/t_integration.log
