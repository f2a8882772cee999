cd $GRAFT_REPO_ROOT
for nb in 8 16 12; do
  echo "LQ_NB=$nb"
  OG_EXTRA_HIPFLAGS="-DOGSQP_LQ_NB=$nb" timeout 900 python tests/perf/solve_timing.py polar_tsto --sqp-core hip 2>&1 | tail -1 | cut -c1-400
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
