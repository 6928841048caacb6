cd $GRAFT_REPO_ROOT
for r in 1 2; do for k in 4 6 8 10 12; do echo -n "[shallow=$k] "; C2A_PEEL_SHALLOW=$k NO_STATS=1 tools/peel_try.sh | cut -c1-40; done; done
for cfg in "C2A_PEEL_SHALLOW=8 C2A_PEEL_SEED_CHUNK=2" "C2A_PEEL_SHALLOW=8 C2A_PEEL_SEED_CHUNK=8" "C2A_PEEL_SHALLOW=8 C2A_PEEL_SEED_CHUNK=1"; do echo -n "[$cfg] "; env $cfg NO_STATS=1 tools/peel_try.sh | cut -c1-40; done
