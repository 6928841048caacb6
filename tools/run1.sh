cd $GRAFT_REPO_ROOT
python tools/locality_ceiling.py > gpurun_out/r04_ceiling.txt 2>&1
NO_STATS=1 tools/peel_try.sh > gpurun_out/r04_base_try.txt 2>&1
cat gpurun_out/r04_ceiling.txt gpurun_out/r04_base_try.txt
