#!/bin/bash
# usage (GPU box): tools/try_new.sh  -- the strip tests on build_ab/sp_new.so, then tools/ab_strip.sh over build_ab/sp_*.so
cd "$GRAFT_REPO_ROOT" || exit 1
L=deepsphere-cosmo-tf2_amd/deepsphere/_lib/libdsphere_hip.so
cp $L /tmp/keep0.so
cp build_ab/sp_new.so $L
timeout 300 python -m pytest tests/test_gpu_round3.py -x -q -m gpu 2>&1 | tail -3
cp /tmp/keep0.so $L
timeout 400 tools/ab_strip.sh 2>&1 | grep -E "==|strip5"
