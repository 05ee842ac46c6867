cd $GRAFT_REPO_ROOT
bash tools/ab_env.sh "ZSG_FRESH_OUT=0" "ZSG_FRESH_OUT=1" 2>&1 | tee gpurun_out/ab_fresh_out.txt
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/gpu_tests_mid2.log
