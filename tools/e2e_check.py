import json, sys, time, os
sys.path.insert(0, os.getcwd())
from mamdr_amd import cli
for cfgp, over in (("config/Taobao-10/deepctr_DN+DR.json", dict(epoch=2, patience=1)),
                   ("config/Taobao-10/star_taobao.json", dict(epoch=2, patience=1)),
                   ("config/Amazon_6/deepfm_DN.json", dict(epoch=1, patience=1))):
    cfg = json.load(open(cfgp))
    cfg["train"].update(over)
    cfg["train"].update(result_save_path="/tmp/res", checkpoint_path="/tmp/ckpt")
    if "Amazon" in cfgp:
        cfg["dataset"]["synthetic_scale"] = 0.05
    t0 = time.time()
    import io, contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        r = cli.main(cfg)
    print(cfgp, cfg["model"]["name"], "avg_loss %.4f avg_auc %.4f" % (r[0], r[1]), "wall %.1f s" % (time.time() - t0))
