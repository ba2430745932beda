# AUC of the Star tower (tensors outside theta / phi carried by TailSync) under 1 / 2 / 4 / 8 ranks sharing one GPU over
# gloo; MAMDR_TAIL_SYNC=sum|mean selects how the ranks' displacements of those tensors are combined (default mean)
mkdir -p gpurun_out/r03u
C=config/Taobao-10/star_taobao.json
MODE=${MAMDR_TAIL_SYNC:-mean}
if [ "$1" = "single" ]; then
for seed in 123 124; do
  timeout 600 python tools/dist_auc.py $C 4 sharded $seed star_meta_mamdr 2>/dev/null | grep DISTAUC | tee -a gpurun_out/r03u/dist_auc_star.jsonl | cut -c1-200
done
fi
for n in 2 4 8; do
  MAMDR_TAIL_SYNC=$MODE MAMDR_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29611 tools/dist_auc.py $C 4 sharded 123 star_meta_mamdr 2>gpurun_out/r03u/err_$n.log | grep DISTAUC | sed "s/DISTAUC {/DISTAUC {\"tail_sync\": \"$MODE\", /" | tee -a gpurun_out/r03u/dist_auc_star.jsonl | cut -c1-200
done
