# A/B of the fused loss kernel: LEC_LIB_PATH = other builds of liblecone.so (VARIANTS: suffixes of learning_embeddings_amd/liblecone_<v>.so; "new" = the current build),
# four sizes, alternating runs (us per launch).
R=$GRAFT_REPO_ROOT
for i in 1 2; do for v in ${VARIANTS:-prev new}; do
  if [ $v = new ]; then unset LEC_LIB_PATH; else export LEC_LIB_PATH=$R/learning_embeddings_amd/liblecone_$v.so; fi
  for shape in "256 5 10 2000" "256 256 10 50000" "4096 256 10 50000" "256 256 128 50000"; do
    echo "$v$i [$shape] $(python3 $R/tools/prof_cone.py $shape 2>/dev/null | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('%.1f us %.0f GB/s' % (d['us'], d['GBps']))")"
  done
done; done
