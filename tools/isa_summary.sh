#!/bin/bash
# usage: tools/isa_summary.sh <file.hip> <kernel-name-substring>  -> instruction-class run-length summary
cd /root/repo/emlight_amd/csrc
rm -rf /tmp/isa && mkdir -p /tmp/isa && cd /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -save-temps -c /root/repo/emlight_amd/csrc/$1 -I/root/repo/emlight_amd/csrc -o /tmp/isa/x.o 2>/dev/null
S=$(ls *gfx950.s | head -1)
awk -v pat="$2" '
  /^_Z.*:/ { inside = (index($0, pat) > 0) }
  inside { print }
  inside && /s_endpgm/ { inside = 0 }' $S > k.s
echo "lines $(wc -l < k.s) mfma $(grep -c v_mfma k.s) vmem_ld $(grep -c -E "global_load|buffer_load" k.s) vmem_st $(grep -c global_store k.s) ds_read $(grep -c ds_read k.s) ds_write $(grep -c ds_write k.s) waitcnt $(grep -c s_waitcnt k.s) scratch $(grep -c scratch_ k.s)"
grep -E "^\s+(s_waitcnt|ds_read|ds_write|ds_bpermute|v_mfma|s_barrier|global_load|global_store|scratch_|s_cbranch|v_accvgpr|s_endpgm)" k.s | awk '{k=$1; if ($1=="s_waitcnt") k=$1" "$2; sub(/_b(32|64|96|128)$/,"",k); sub(/_dword.*/,"",k); sub(/v_mfma.*/,"MFMA",k); print k}' | uniq -c | awk '{printf "%s %s%s; ", $1, $2, ($3?" "$3:"")}' | fold -w 220
echo
