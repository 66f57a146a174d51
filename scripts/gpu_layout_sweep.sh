#!/bin/bash
# lanes-per-row sweep: every K with 8 / 16 / 32 lanes per row (variants built by scripts/build_variant.sh)
cd $GRAFT_REPO_ROOT
cp transductive-clip_amd/tclip_amd/libtclip.so /tmp/libtclip_orig.so
t() { python scripts/prof_small.py $1 10 100 20 2>&1 | tail -1 | sed 's/ mm_ms.*updates\/s=/ U\/s=/; s/ mm_iters.*//'; }
for K in 10 37 47 100 196; do
  cp gpurun_variants/g16.so transductive-clip_amd/tclip_amd/libtclip.so; echo "G8(<=64)/G16: $(t $K)"
  cp gpurun_variants/g16all.so transductive-clip_amd/tclip_amd/libtclip.so; echo "G16 all:      $(t $K)"
  echo "G32:          $(TCLIP_WIDE=1 t $K)"
done
cp /tmp/libtclip_orig.so transductive-clip_amd/tclip_amd/libtclip.so
