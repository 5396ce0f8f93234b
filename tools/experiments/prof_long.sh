# rocprofv3 kernel trace of the long-chain diagnostics at two shapes (round 6): which kernel takes what
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shp in "16384 20000 3" "65536 8000 3"; do
  rm -rf /tmp/pl
  timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pl -o pl --output-format csv -- python3 $R/tools/experiments/prof_long.py $shp > /tmp/pl.log 2>&1 < /dev/null
  echo "== $shp"
  for f in /tmp/pl/*/*kernel_stats.csv /tmp/pl/*kernel_stats.csv; do [ -f "$f" ] && head -8 "$f" | cut -c1-220; done
done
