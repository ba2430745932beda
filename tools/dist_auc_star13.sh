# BASELINE configs[4]'s model (Star tower, MAMDR, trainable Amazon-13 tables; 10 % of the rows, 3 epochs) under 1 / 2 / 4 ranks
mkdir -p gpurun_out/r03w
C=config/Amazon_13/star_DN+DR.json
MAMDR_DIST_AUC_SCALE=0.1 timeout 1200 python tools/dist_auc.py $C 3 sharded 123 star_meta_mamdr 2>gpurun_out/r03w/err_1.log | grep DISTAUC | tee -a gpurun_out/r03w/dist_auc_star13.jsonl | cut -c1-160
MAMDR_DIST_AUC_SCALE=0.1 timeout 1200 python tools/dist_auc.py $C 3 sharded 124 star_meta_mamdr 2>gpurun_out/r03w/err_1b.log | grep DISTAUC | tee -a gpurun_out/r03w/dist_auc_star13.jsonl | cut -c1-160
for n in 2 4; do
  MAMDR_DIST_AUC_SCALE=0.1 MAMDR_SHARE_GPU=1 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29631 tools/dist_auc.py $C 3 sharded 123 star_meta_mamdr 2>gpurun_out/r03w/err_$n.log | grep DISTAUC | tee -a gpurun_out/r03w/dist_auc_star13.jsonl | cut -c1-160
done
tail -3 gpurun_out/r03w/err_4.log | cut -c1-200
