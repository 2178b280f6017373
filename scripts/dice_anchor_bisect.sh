#!/bin/bash
# which switch moves the float32 Adam loss curve away from the oracle's?  (tests/test_hip_dice.py anchored test, env toggles)
run() { echo "== $*"; env "$@" python -m pytest tests/test_hip_dice.py -x -q -m gpu -k anchored 2>&1 | grep -E "AssertionError: \(array|passed|failed" | head -3; }
run A=1
run SAUNET_FUSED_BASIC_BLOCK=0
run SAUNET_DENSE_BNPRO=0
run SAUNET_DENSE_COEFF_CORRECT=0
run SAUNET_DENSE_WGRAD_GROUPED=0
run SAUNET_DENSE_DGRAD=0
run SAUNET_FUSED_BASIC_BLOCK=0 SAUNET_DENSE_BNPRO=0 SAUNET_DENSE_COEFF_CORRECT=0 SAUNET_DENSE_WGRAD_GROUPED=0 SAUNET_DENSE_DGRAD=0
