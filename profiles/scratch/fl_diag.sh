cd "$GRAFT_REPO_ROOT"
make -s -C svgrasterize.py_amd/csrc clean && make -s -C svgrasterize.py_amd/csrc EXTRA="-DSVGR_DBG_FL_NOATOMIC" 2>/dev/null
bash profiles/trace_workload.sh tiger2048 2>/dev/null | grep flatten
make -s -C svgrasterize.py_amd/csrc clean && make -s -C svgrasterize.py_amd/csrc
