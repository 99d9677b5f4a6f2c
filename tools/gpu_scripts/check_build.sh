#!/bin/bash
# every GPU script starts here: refuse to measure a library that was not built from the sources in this snapshot
python - <<'PY' || exit 9
import sys, os
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd import build as b
sf = b.LIB + ".stamp"
ok = os.path.exists(b.LIB) and os.path.exists(sf) and open(sf).read() == b._stamp()
print("[check_build] library is", "current" if ok else "STALE - rebuild before measuring")
sys.exit(0 if ok else 1)
PY
