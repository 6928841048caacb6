for w in 8 16 32; do
  C2A_TOURNEY_EXP=$w timeout 300 python bench.py --steps 2 --warmup 1 --cpu-sample-layers 0 --no-width64 --no-artefacts --check 2>&1 | grep -E "tourney exp|checked|Error|error" | cut -c1-400 | sed 's/.*"checked"/checked/'
done
