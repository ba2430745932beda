mkdir -p gpurun_out/r03v
for seed in 124 125; do
MAMDR_DIST_AUC_SCALE=0.2 timeout 900 python tools/dist_auc.py config/Amazon_6/deepfm_DN.json 4 sharded $seed deepfm_meta_domain_negotiation 2>/dev/null | grep DISTAUC | tee -a gpurun_out/r03v/dist_auc_seeds.jsonl | cut -c1-150
timeout 900 python tools/dist_auc.py config/Taobao-10/deepctr_DN_taobao_10.json 6 sharded $seed mlp_meta_domain_negotiation 2>/dev/null | grep DISTAUC | tee -a gpurun_out/r03v/dist_auc_seeds.jsonl | cut -c1-150
timeout 900 python tools/dist_auc.py config/Taobao-10/deepctr_reptile_taobao_10.json 6 sharded $seed mlp_meta_reptile 2>/dev/null | grep DISTAUC | tee -a gpurun_out/r03v/dist_auc_seeds.jsonl | cut -c1-150
done
