bash scripts/gpu_r6_train_ab.sh "new:f16x3_bwd2 dst:f16x3_bwd2 nost:f16x3_bwd2 new:f16x3_bwd1 dst:f16x3_bwd1 nost:f16x3_bwd1" pytest
TRAIN_PREC=f16x3_bwd2 TAG=_bwd2 bash scripts/pmc_train_sq.sh 2>&1 | grep -A12 "chain_bwd_h\|mlp_f16x3\|wgrad_jobs" | head -120
TRAIN_PREC=f16x3_bwd1 TAG=_bwd1 bash scripts/pmc_train_sq.sh 2>&1 | grep -A12 "chain_bwd_h" | head -60
