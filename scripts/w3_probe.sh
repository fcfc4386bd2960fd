#!/bin/bash
# timing experiments on the fixed-geometry weight-gradient kernel: which phase the time sits in (PCUDA_W3DBG bits:
# 1 no X loads, 2 no dZ loads, 4 no MFMA phase, 8 no commit)
cd "${GRAFT_REPO_ROOT:-.}"
for d in ${DBGS:-0 1 2 3 4 8 12 7 15}; do
  echo "PCUDA_W3DBG=$d: $(PCUDA_W3DBG=$d python scripts/conv_micro.py ${CASES:-g64} 2>&1 | tail -1 | sed 's/.*wgrad/wgrad/')"
done
