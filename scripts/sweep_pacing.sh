#!/bin/bash
# dev tool: pacing knobs (probe builds read them from the environment) x library variants
SPECS=${SPECS:-"64x31_noquote:8,16x32_noquote:1,1024x4_dense:1"}
for d in ${DELAYS:-0 8 16}; do for p in ${PRIOS:-0 1}; do echo "delay=$d prio=$p"; CSVSIMD_PROBE_MODE=0 CSVSIMD_PROBE_EMIT_DELAY=$d CSVSIMD_PROBE_COUNT_PRIO=$p python scripts/ab_variants.py "$SPECS" 1; done; done
