#!/bin/bash
# the bench lines of the evidence set alone (after a bench.py-only change)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; bash tools/gpu_r06_evidence.sh final "smoke bench"
