cd $GRAFT_REPO_ROOT
NO_STATS=1 tools/peel_try.sh > gpurun_out/r04_try.txt 2>&1
python tools/locality_ceiling.py >> gpurun_out/r04_try.txt 2>&1
cat gpurun_out/r04_try.txt
