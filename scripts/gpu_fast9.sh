#!/bin/bash
# FAST-9 alone on 32 4K frames: the default build and every library under scripts/variants/
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
python scripts/fast9_time.py 2>/dev/null | grep fast9_ms
for f in scripts/variants/*.so; do [ -f "$f" ] && VARIANT_LIB=$f python scripts/fast9_time.py 2>/dev/null | grep fast9_ms; done
python scripts/fast9_time.py 2>/dev/null | grep fast9_ms
