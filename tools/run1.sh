cd $GRAFT_REPO_ROOT
bash tools/ab.sh 3 base hoist 2>&1 | cut -c1-50
