"""Chain-parallel multi-GPU driver: one process per GPU, one independent chain per rank,
no data-path collective; exactly one gather of the chain histories at the end (RCCL over xGMI when the
process group's backend is ``nccl``; ``gloo`` on CPU tensors in the tests)."""
import os

import numpy as np
import torch
import torch.distributed as dist


def select_device():
    """One process per GPU: bind this process to cuda:LOCAL_RANK BEFORE any process group exists (RCCL places its
    communicator on the current device)."""
    if not torch.cuda.is_available():
        return None
    local = int(os.environ.get('LOCAL_RANK', os.environ.get('RANK', '0')))
    torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    return torch.device('cuda', torch.cuda.current_device())


def ranks_share_a_card():
    """True when this node runs more ranks than it has cards (a rehearsal on a one-GPU box).  Decided from
    LOCAL_WORLD_SIZE only -- torchrun and ``bench.py --gpus N`` export it; without it nothing is guessed (WORLD_SIZE
    counts the ranks of every node)."""
    lws = os.environ.get('LOCAL_WORLD_SIZE')
    return lws is not None and torch.cuda.is_available() and int(lws) > torch.cuda.device_count()


def init_distributed(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torchrun); no-op for a single process --
    unless GPMC_FORCE_DIST=1, which creates the group for one rank too, so that the RCCL code path (communicator on
    the bound device, device barrier, CUDA-tensor collectives) can be executed on a one-GPU box."""
    dev = select_device()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    force = os.environ.get('GPMC_FORCE_DIST') == '1'
    if world == 1 and not force:
        return 0, 1
    if dev is not None and ranks_share_a_card():
        # several ranks share one card: a look-ahead panel launch gated on the flags of a GEMM that has not been
        # dispatched yet holds CU slots while it waits, and two processes' launches together can hold all of them
        # (info = -1001 in round 2).  shared_card makes the hand-over stream-ordered instead (csrc/potrf.hip); a rank
        # with a card to itself keeps the in-launch hand-over.
        from . import _lib
        _lib.set_option('shared_card', 1)
    if not dist.is_initialized():
        if backend is None:   # GPMC_DIST_BACKEND=gloo: rehearse several ranks on one GPU (RCCL wants one device per rank)
            backend = os.environ.get('GPMC_DIST_BACKEND') or ('nccl' if dev is not None else 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if backend == 'nccl':
            dist.init_process_group(backend=backend, device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size()


def _forced():
    return os.environ.get('GPMC_FORCE_DIST') == '1'


def gather_histories(histHyp, histF=None, device=None):
    """all_gather per-rank histories -> (world, 3, iters) [, (world, n, iters)] ndarrays on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not _forced()):
        return (np.asarray(histHyp)[None], None if histF is None else np.asarray(histF)[None])
    world = dist.get_world_size()
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else 'cpu'
    out = []
    for h in (histHyp, histF):
        if h is None:
            out.append(None)
            continue
        t = torch.as_tensor(np.ascontiguousarray(h), dtype=torch.float64).to(device)
        bufs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(bufs, t)
        out.append(torch.stack(bufs).cpu().numpy())
    return out[0], out[1]


def gather_vector(v):
    """Each rank holds a float vector with NaN where it has no result; one all_gather, then the non-NaN entry per
    slot (every slot is owned by exactly one rank) -> the full vector on every rank."""
    v = np.asarray(v, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not _forced()):
        return v
    device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.as_tensor(v).to(device)
    bufs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(bufs, t)
    allv = torch.stack(bufs).cpu().numpy()
    have = ~np.isnan(allv)
    if np.any(have.sum(axis=0) > 1):
        raise RuntimeError('gather_vector: a slot was filled by more than one rank')
    return np.where(have.any(axis=0), np.nansum(allv, axis=0), np.nan)


def run_chains(x, y, iters, hyp0, seed_base=124, gather_f=False, chain_fn=None):
    """Rank r runs chain r (legacy np.random seeded with seed_base + r) on its own GPU, then one gather."""
    rank, world = init_distributed()
    np.random.seed(seed_base + rank)
    if chain_fn is None:
        from . import framework
        chain_fn = framework.run_chain
    histF, histHyp, ntries = chain_fn(x, y, iters, hyp0)
    allHyp, allF = gather_histories(histHyp, histF if gather_f else None)
    return dict(rank=rank, world=world, histHyp=allHyp, histF=allF, proposals=int(np.sum(ntries)))
