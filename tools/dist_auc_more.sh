# AUC under 1 / 2 / 4 ranks (sharing one GPU over gloo) of the other sharded wrappers: Domain Negotiation on the DeepFM tower
# with trainable tables (BASELINE configs[2]'s model, Amazon-6 tables, 20 % of the rows) and Reptile on the mlp tower
mkdir -p gpurun_out/r03v
run() { # n config epochs name
  if [ $1 = 1 ]; then
    timeout 900 python tools/dist_auc.py $2 $3 sharded 123 $4 2>gpurun_out/r03v/err_$4_1.log | grep DISTAUC
  else
    MAMDR_SHARE_GPU=1 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port 29621 tools/dist_auc.py $2 $3 sharded 123 $4 2>gpurun_out/r03v/err_$4_$1.log | grep DISTAUC
  fi | tee -a gpurun_out/r03v/dist_auc_more.jsonl | cut -c1-190
}
for n in 1 2 4; do MAMDR_DIST_AUC_SCALE=0.2 run $n config/Amazon_6/deepfm_DN.json 4 deepfm_meta_domain_negotiation; done
for n in 1 2 4; do run $n config/Taobao-10/deepctr_reptile_taobao_10.json 6 mlp_meta_reptile; done
for n in 1 2 4; do run $n config/Taobao-10/deepctr_DN_taobao_10.json 6 mlp_meta_domain_negotiation; done
