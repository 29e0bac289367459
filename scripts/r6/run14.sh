#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
L=ram-dsir_amd/ramdsir/libramdsir_hip_dbg.so
echo "== product flags"; bash scripts/ram_prof.sh u8 400 7 2>&1 | tail -4
bash scripts/r6/ab_many.sh 3 ab/product.so ab/noslp_all.so
