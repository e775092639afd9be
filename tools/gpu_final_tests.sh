#!/bin/bash
# the GPU test suite + smoke of the evidence set alone
cd "${GRAFT_REPO_ROOT:-/root/repo}"; bash tools/gpu_r06_evidence.sh final "test smoke"
