# Sourced by the A/B scripts whose builds select an experiment branch (-DMPG_AB_*, -DMPG_TR_IMAGE, -DMPG_IMG_LAYOUT=n): since round 6 the
# shipped sources hold the shipped expansion only (tools/strip_ablation.py; ISA unchanged, tools/isa_hash.py) and the experiment branches
# live in archive/proto/ablation_macros.patch, cut against commit 1472a68.  This puts them back into the WORKING TREE (never commit that);
# `git checkout mpg_amd/csrc` removes them again.
if ! grep -q "MPG_AB_\|MPG_TR_IMAGE" mpg_amd/csrc/mlp_core.h; then
    git apply archive/proto/ablation_macros.patch 2>/dev/null || git apply --3way archive/proto/ablation_macros.patch || {
        echo "archive/proto/ablation_macros.patch does not apply to this tree (cut against 1472a68): check that commit out, or port the branch you need" >&2
        exit 1
    }
    echo "[ab_scaffold] experiment branches applied to the working tree (git checkout mpg_amd/csrc to drop them)" >&2
fi
