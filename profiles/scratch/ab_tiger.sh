cd "$GRAFT_REPO_ROOT"
for v in "coop:" "nocoop:-DSVGR_DBG_NO_COOP_SLABS" "coop2:" "nocoop2:-DSVGR_DBG_NO_COOP_SLABS"; do
  name="${v%%:*}"; flags="${v#*:}"
  make -s -C svgrasterize.py_amd/csrc clean && make -s -C svgrasterize.py_amd/csrc EXTRA="$flags" 2>/dev/null
  echo "== $name"; bash profiles/trace_workload.sh tiger2048 2>/dev/null | head -5
done
make -s -C svgrasterize.py_amd/csrc clean && make -s -C svgrasterize.py_amd/csrc
