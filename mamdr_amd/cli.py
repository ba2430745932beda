"""`run.py --config <json>` surface and the name registry (mirror of the reference's run.py).

Dispatch is by substring of config.model.name, in the reference's order (run.py:37-85):
  tower    'star' -> Star | any of mlp,wdl,nfm,autoint,ccpm,pnn,deepfm -> DeepCTR |
           shared_bottom,mmoe,ple -> DeepMTLCTR
  wrappers 'uncertainty_weight', 'pcgrad', then 'meta' + (domain_negotiation | mamdr |
           reptile | mldg | <else> MAML)
  modes    'separate' -> per-domain training; otherwise train() + val_and_test("test");
           'finetune' -> load best + separate_train_val_test(init_parms=False)
Every tower name of the reference's registries is built (run.py:37-47; deepctr.py:24-50: mlp wdl nfm autoint ccpm pnn
deepfm; deep_mtl_ctr.py:25-49: shared_bottom mmoe ple; star); what is not (Star's auxiliary net and BatchNormalization form, ple with more than one
level, uncertainty weighting on the multi-task towers) raises NotImplementedError naming why.
"""
import argparse
import json

DEEP_CTR_LIST = ["mlp", "wdl", "nfm", "autoint", "ccpm", "pnn", "deepfm"]
MTL_DEEP_CTR_LIST = ["shared_bottom", "mmoe", "ple"]


def in_name_list(x, name_list):
    return any(n in x for n in name_list)


def build_model(config, dataset, engine_factory=None):
    from .model_zoo import MAML, MAMDR, MLDG, DeepCTR, DomainNegotiation, PCGrad, Reptile, Star, UncertaintyWeight
    name = config["model"]["name"]
    if "star" in name:
        model = Star(dataset, config, engine_factory)
    elif in_name_list(name, DEEP_CTR_LIST):
        model = DeepCTR(dataset, config, engine_factory)
    elif in_name_list(name, MTL_DEEP_CTR_LIST):
        from .model_zoo import DeepMTLCTR
        model = DeepMTLCTR(dataset, config, engine_factory)
    else:
        raise ValueError("model: {} not found".format(name))
    if "uncertainty_weight" in name:
        model = UncertaintyWeight(model)
    if "pcgrad" in name:
        model = PCGrad(model)
    if "meta" in name:
        if "domain_negotiation" in name:
            model = DomainNegotiation(model)
        elif "mamdr" in name:
            model = MAMDR(model)
        elif "reptile" in name:
            model = Reptile(model)
        elif "mldg" in name:
            model = MLDG(model)
        else:
            model = MAML(model)
    return model


def init_distributed():
    """one process per GPU when launched under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the
    environment): RCCL process group; MAMDR_SHARE_GPU=1 (testing on a 1-GPU box) keeps every rank on device 0
    over gloo.  Returns (rank, world_size)."""
    import os
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws <= 1:
        return 0, 1
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        # an explicit timeout: a rank that hangs in the preflight's ring or in a collective is cut after
        # minutes, not after the backend's default of 10 - 30; MAMDR_COMM_TIMEOUT (seconds) overrides
        import datetime
        timeout = datetime.timedelta(seconds=float(os.environ.get("MAMDR_COMM_TIMEOUT", "300")))
        if os.environ.get("MAMDR_SHARE_GPU") == "1":
            if torch.cuda.is_available():
                torch.cuda.set_device(0)
            dist.init_process_group("gloo", timeout=timeout)
        else:
            # one process per GPU OF THIS NODE: LOCAL_WORLD_SIZE ranks share the node's devices (WORLD_SIZE counts the
            # ranks of every node)
            local_ws = int(os.environ.get("LOCAL_WORLD_SIZE", str(local + 1)))
            if torch.cuda.device_count() < max(local_ws, local + 1):
                raise RuntimeError("%d local ranks but %d visible GPUs (one process per GPU)" % (
                    max(local_ws, local + 1), torch.cuda.device_count()))
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", timeout=timeout)
        # first contact with the communicator: values of an all-reduce, a send / recv ring and a broadcast checked on
        # every rank; a failing ring switches the phi hand-over to broadcasts (parallel.preflight)
        from . import parallel
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        parallel.preflight(dev)
    return dist.get_rank(), dist.get_world_size()


def n_lanes(config):
    """train.lanes (not a key of the reference's configs; MAMDR_LANES overrides; default 1 = the reference's one
    sequential chain): the sharded epoch of SURVEY 8e run by that many LANES of this process on one GPU
    (parallel.LaneGroup) -- a lane run is the run of as many ranks, with the lanes' kernels overlapping on the device."""
    import os
    return max(1, int(os.environ.get("MAMDR_LANES") or config["train"].get("lanes", 1) or 1))


def main(config, engine_factory=None, on_model=None):
    """run.py:71-89.  on_model(model) (optional) sees the built, wrapped model before training starts (tests attach
    their recorders there)."""
    from .utils import MultiDomainDataset
    rank, world = init_distributed()
    name = config["model"]["name"]
    lanes = n_lanes(config)
    if (world > 1 or lanes > 1) and not ("meta" in name and ("mamdr" in name or "domain_negotiation" in name or "reptile" in name)):
        raise NotImplementedError("multi-process / multi-lane runs shard the MAMDR (DN + DR), Domain Negotiation and Reptile "
                                  "wrappers only; got '%s'" % name)
    dataset = MultiDomainDataset(config["dataset"])
    if lanes > 1:
        # (under N processes the lanes of every rank are a slice of ONE world of N * lanes participants: a rank's DR queries
        # and DN sub-sequence are dealt on to its lanes, collectives = lane step + one inter-rank collective per process --
        # parallel.py, "RANKS x LANES")
        from . import parallel
        import os
        import torch
        if not torch.cuda.is_initialized():
            # every lane's stream on a hardware queue of its own (the HIP runtime reads this when it initialises; its
            # default of 4 queues is shared with torch's other streams: profiles/r05_lanes_bench.txt)
            os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(8, 2 * lanes)))
        # (the dataset is read-only: one copy for all lanes; every lane builds its own model = its own engine and stream)
        # (train.lane_sum_block: a test knob -- a one-process run of N * L lanes that adds up in the order of N processes of L
        # lanes, the bit-for-bit reference of the composed run: tests/test_abi_and_parallel.py)
        return parallel.LaneGroup(lanes, outer=(rank, world), sum_block=config["train"].get("lane_sum_block")).run(
            lambda lane: _run(config, dataset, engine_factory, on_model, rank * lanes + lane))[0]
    return _run(config, dataset, engine_factory, on_model, rank)


def _run(config, dataset, engine_factory, on_model, rank):
    name = config["model"]["name"]
    model = build_model(config, dataset, engine_factory)
    if on_model is not None:
        on_model(model)
    if "separate" in name:
        avg_loss, avg_auc, domain_loss, domain_auc = model.separate_train_val_test()
    else:
        model.train()
        print("Test Result: ")
        avg_loss, avg_auc, domain_loss, domain_auc = model.val_and_test("test")
    if "finetune" in name:
        model.load_model(model.checkpoint_path)      # the best checkpoint (kept on the device as well)
        print("Finetune: ")
        avg_loss, avg_auc, domain_loss, domain_auc = model.separate_train_val_test(init_parms=False)
    if rank == 0:
        model.save_result(avg_loss, avg_auc, domain_loss, domain_auc)
    return avg_loss, avg_auc, domain_loss, domain_auc


def cli(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", type=str, help="Train config file", required=True)
    # (not an option of the reference's run.py: the sharded epoch on one GPU, see n_lanes)
    parser.add_argument("--lanes", type=int, default=None, help="run the sharded epoch on this many lanes of one process (train.lanes)")
    args = parser.parse_args(argv)
    with open(args.config, "r") as f:
        config = json.load(f)
    if args.lanes is not None:
        config["train"]["lanes"] = args.lanes
    return main(config)
