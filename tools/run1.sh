cd $GRAFT_REPO_ROOT
bash tools/ab.sh 3 base p16 p4 2>&1 | cut -c1-50
