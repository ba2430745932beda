# the *_finetune names (per-domain SGD stage after training: owners finetune their domains) under 1 / 2 / 4 ranks
mkdir -p gpurun_out/r03x
run() { # n config epochs name
  if [ $1 = 1 ]; then
    timeout 900 python tools/dist_auc.py $2 $3 sharded 123 $4 2>gpurun_out/r03x/err_$4_1.log | grep DISTAUC
  else
    MAMDR_SHARE_GPU=1 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port 29641 tools/dist_auc.py $2 $3 sharded 123 $4 2>gpurun_out/r03x/err_$4_$1.log | grep DISTAUC
  fi | tee -a gpurun_out/r03x/dist_auc_finetune.jsonl | cut -c1-170
}
for n in 1 2 4; do run $n config/Taobao-10/deepctr_DN+DR.json 4 mlp_meta_mamdr_finetune; done
for n in 1 2 4; do run $n config/Taobao-10/star_taobao.json 3 star_meta_mamdr_finetune; done
