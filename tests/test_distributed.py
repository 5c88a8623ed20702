"""Multi-rank path on CPU: world_size-2 gloo.  Each rank solves its contiguous
shard (the CPU oracle stands in for the HIP launch, which needs a GPU) and the
joint velocities are all-gathered; the result must equal the single-process
answer on the full batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from casclik_amd import skills
from casclik_amd.distributed import shard_bounds, all_gather_rows


def test_shard_bounds_partition():
    for B in (0, 1, 7, 64, 16384, 131072 + 3):
        for G in (1, 2, 3, 8):
            cuts = [shard_bounds(B, r, G) for r in range(G)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B
            assert all(cuts[r][1] == cuts[r + 1][0] for r in range(G - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, B, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import clik_oracle
        fk = skills.iiwa()
        spec = skills.stack_skill(fk)
        Q, Y = skills.synthetic_inputs(fk, B, seed=4, distribution="mixed")
        lo, hi = shard_bounds(B, rank, world)
        dq, mode = clik_oracle.pinv_solve_batch(spec, skills.STACK_OPTIONS, 0.0, Q[lo:hi], Y=Y[lo:hi])
        full = all_gather_rows(torch.from_numpy(dq), n_rows_total=B)
        modes = all_gather_rows(torch.from_numpy(mode.astype(np.int64)).reshape(-1, 1), n_rows_total=B)
        np.save(os.path.join(out_dir, "dq_%d.npy" % rank), full.numpy())
        np.save(os.path.join(out_dir, "mode_%d.npy" % rank), modes.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [24, 25])      # even and uneven shards
def test_two_rank_gloo_matches_single_process(tmp_path, B):
    from oracle import clik_oracle
    port = _free_port()
    mp.spawn(_worker, args=(2, port, B, str(tmp_path)), nprocs=2, join=True)
    fk = skills.iiwa()
    Q, Y = skills.synthetic_inputs(fk, B, seed=4, distribution="mixed")
    ref, rmode = clik_oracle.pinv_solve_batch(skills.stack_skill(fk), skills.STACK_OPTIONS, 0.0, Q, Y=Y)
    for rank in range(2):
        got = np.load(os.path.join(str(tmp_path), "dq_%d.npy" % rank))
        gm = np.load(os.path.join(str(tmp_path), "mode_%d.npy" % rank))[:, 0]
        assert got.shape == ref.shape
        assert np.array_equal(got, ref) and np.array_equal(gm, rmode)


class _OracleController(object):
    """stands in for a device controller in the CPU test of ShardedController (same solve_batch shape)"""

    def __init__(self, spec, options):
        self.spec, self.options = spec, options

    def solve_batch(self, time_var, robot_var, input_var=None):
        from oracle import clik_oracle
        dq, mode = clik_oracle.pinv_solve_batch(self.spec, self.options, time_var, robot_var, Y=input_var)
        return dq, None, mode


def _sharded_worker(rank, world, port, B, out_dir, device):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from casclik_amd.distributed import ShardedController
        fk = skills.iiwa()
        spec = skills.stack_skill(fk)
        Q, Y = skills.synthetic_inputs(fk, B, seed=4, distribution="mixed")
        lo, hi = shard_bounds(B, rank, world)
        if device == "cpu":
            inner = _OracleController(spec, skills.STACK_OPTIONS)
            q_in, y_in = Q[lo:hi], Y[lo:hi]
        else:
            import casclik_amd as cc
            inner = cc.PseudoInverseController(skill_spec=spec, options=dict(skills.STACK_OPTIONS))
            inner.setup_problem_functions()
            q_in, y_in = torch.from_numpy(Q[lo:hi]).cuda(), torch.from_numpy(Y[lo:hi]).cuda()
        sc = ShardedController(inner)
        local = sc.solve_batch(0.0, q_in, input_var=y_in)
        assert local.shape[0] == hi - lo
        full = sc.solve_batch(0.0, q_in, input_var=y_in, gather=True, n_rows_total=B)
        np.save(os.path.join(out_dir, "sharded_%d.npy" % rank), full.cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [24, 25])
def test_sharded_controller_two_ranks_gloo(tmp_path, B):
    from oracle import clik_oracle
    mp.spawn(_sharded_worker, args=(2, _free_port(), B, str(tmp_path), "cpu"), nprocs=2, join=True)
    fk = skills.iiwa()
    Q, Y = skills.synthetic_inputs(fk, B, seed=4, distribution="mixed")
    ref, _ = clik_oracle.pinv_solve_batch(skills.stack_skill(fk), skills.STACK_OPTIONS, 0.0, Q, Y=Y)
    for rank in range(2):
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "sharded_%d.npy" % rank)), ref)


@pytest.mark.gpu
def test_sharded_hip_launch_two_ranks_on_one_gpu(tmp_path):
    """Two ranks, each running the HIP controller on ITS shard (both on cuda:0: this box has one GPU), rows
    gathered over gloo: equals the single-process answer on the full batch bit for bit."""
    import casclik_amd as cc
    B = 1000
    mp.spawn(_sharded_worker, args=(2, _free_port(), B, str(tmp_path), "cuda"), nprocs=2, join=True)
    fk = skills.iiwa()
    Q, Y = skills.synthetic_inputs(fk, B, seed=4, distribution="mixed")
    ctrl = cc.PseudoInverseController(skill_spec=skills.stack_skill(fk), options=dict(skills.STACK_OPTIONS))
    ctrl.setup_problem_functions()
    ref = ctrl.solve_batch(0.0, Q, input_var=Y)[0]
    for rank in range(2):
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "sharded_%d.npy" % rank)), ref)


def _nccl_worker(rank, world, port, B, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        import casclik_amd as cc
        from casclik_amd.distributed import ShardedController
        fk = skills.iiwa()
        Q, Y = skills.synthetic_inputs(fk, B, seed=4, distribution="mixed")
        lo, hi = shard_bounds(B, rank, world)
        inner = cc.PseudoInverseController(skill_spec=skills.stack_skill(fk),
                                           options=dict(skills.STACK_OPTIONS, device="cuda:%d" % rank))
        inner.setup_problem_functions()
        sc = ShardedController(inner)
        full = sc.solve_batch(0.0, torch.from_numpy(Q[lo:hi]).cuda(), input_var=torch.from_numpy(Y[lo:hi]).cuda(),
                              gather=True, n_rows_total=B)
        np.save(os.path.join(out_dir, "nccl_%d.npy" % rank), full.cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_hip_launch_two_ranks_rccl(tmp_path):
    """One rank per GPU, rows all-gathered by RCCL (backend "nccl"); needs two devices."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs for an RCCL all-gather (this box shows %d)" % torch.cuda.device_count())
    import casclik_amd as cc
    B = 4096
    mp.spawn(_nccl_worker, args=(2, _free_port(), B, str(tmp_path)), nprocs=2, join=True)
    fk = skills.iiwa()
    Q, Y = skills.synthetic_inputs(fk, B, seed=4, distribution="mixed")
    ctrl = cc.PseudoInverseController(skill_spec=skills.stack_skill(fk), options=dict(skills.STACK_OPTIONS))
    ctrl.setup_problem_functions()
    ref = ctrl.solve_batch(0.0, Q, input_var=Y)[0]
    for rank in range(2):
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "nccl_%d.npy" % rank)), ref)
