for dbg in 0 1 2 4 7; do
  echo "== debug mask $dbg"
  SWS_HIP_TILE_DEBUG=$dbg bash tools/pmc_run.sh pmcph_$dbg c3b "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" 2>&1 | grep -A7 "tile_dot2<true, false>" | grep -E "INSTS|WAVES"
done
