# A/B of the fused loss kernel: LEC_LIB_PATH = a previous build of liblecone.so against the current one, four sizes, alternating runs (us per launch).
R=$GRAFT_REPO_ROOT
for i in 1 2; do for v in prev new; do
  if [ $v = prev ]; then export LEC_LIB_PATH=$R/learning_embeddings_amd/liblecone_prev.so; else unset LEC_LIB_PATH; fi
  for shape in "256 5 10 2000" "256 256 10 50000" "4096 256 10 50000" "256 256 128 50000"; do
    echo "$v$i [$shape] $(python3 $R/tools/prof_cone.py $shape 2>/dev/null | tail -1 | cut -c1-300)"
  done
done; done
