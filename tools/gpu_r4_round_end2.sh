#!/bin/bash
# Round 4, after the last kernel change: VAE passes re-taken, the two committed bench lines
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
tools/gpu_r4_profiles.sh vae_b512 2>&1 | grep "rc="
tools/gpu_r4_final.sh 2>&1 | tail -19
