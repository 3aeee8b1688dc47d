#!/bin/bash
# round 5, call F: same-box A/B at 4K of (1) K3 pass 0 with the unreachable corners of its staged rectangle shaved (three workgroups per CU instead of two;
# in-tree) against the whole rectangle (noshave); (2) K1: the hit tests inside the exact-texel exec region (hir), wave-uniform march loops (wave), both,
# the power-of-two table pitch (p2: v_bitop3 cell address), p2 + hir.  Every variant prints the sha1 of its outputs: all must equal the in-tree build's.
O=gpurun_out/r05_f; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 500 bash tools/time_variants.sh ) > $O/variants.txt 2>&1
grep "==\|^K1 ssgi\|^K3\|^frame\|sha1" $O/variants.txt
