for v in 0 1; do echo "variant=$v"; CSVSIMD_PROBE_MODE=0 CSVSIMD_PROBE_VARIANT=$v python scripts/ab_variants.py "64x31_noquote:8,64x31_noquote:2,16x32_noquote:1,16x32_q10:1,1024x4_dense:1" 1; done
