"""CPU-only: the product's kernel source executed under the lane-thread emulation harness
(tests/hostemu) against the fp64 oracle.  This exercises the kernel LOGIC without a GPU; the parity
claims proper are made by tests/test_gpu_parity.py on MI355X with the same scenarios."""
import os

import pytest

from tests import parity_cases as pc
from tests.simharness import ArraySim


# Emulated runs of code that the GPU tests cover at full size (tests/test_gpu_parity.py) and that cost minutes under the lane-thread
# emulator: run them with SO101_SLOW_TESTS=1; the default CPU suite keeps the forward stages, the control step, the env semantics and the default step path against the fused one.
SLOW = pytest.mark.skipif(not os.environ.get("SO101_SLOW_TESTS"), reason="emulated run of a path the GPU tests cover; set SO101_SLOW_TESTS=1")

@pytest.fixture(scope="module")
def make_sim(blobs):
    def f(n, seed=0, **cfg):
        return ArraySim(blobs["f32"], n, backend="emu", seed=seed, **cfg)
    return f


def test_forward_stages(make_sim, blobs):
    pc.check_forward_stages(make_sim, blobs, n=2)


def test_kat1_through_the_kernels(make_sim, blobs, golden):
    pc.check_kat1(make_sim, blobs, golden)


def test_one_control_step(make_sim, blobs):
    pc.check_control_step(make_sim, blobs, n=2, iterations=20)


def test_reward_bitexact(make_sim, blobs):
    pc.check_reward_bitexact(make_sim, blobs, n=24)


def test_env_semantics(make_sim, blobs):
    pc.check_env_semantics(make_sim, blobs, n=1, settle=6, steps=4, last_step=3, iterations=10, eject_substeps=0)     # full delay-line wrap: GPU suite


def test_probe_outlier_state_of_round_6(make_sim, blobs, golden):
    pc.check_probe_outliers(make_sim, blobs, golden, which=[1])


@SLOW
def test_contact_capacity_rule_is_mirrored_by_the_oracle(make_sim, blobs):
    pc.check_contact_capacity_mirrored(make_sim, blobs)


@SLOW
def test_divergence_handling(make_sim, blobs):
    pc.check_divergence_handling(make_sim, blobs)


def test_contact_rich_states(make_sim, blobs, golden):
    pc.check_contact_rich(make_sim, blobs, golden, count=4)


@SLOW
def test_reset_prefetch_is_bit_identical(make_sim):
    pc.check_prefetch_identical(make_sim, n=1, settle=4, steps=4, last_step=1)


@SLOW
def test_settled_store_is_bit_identical(make_sim):
    pc.check_settled_store_identical(make_sim, n=1, settle=4, steps=5, last_step=1, first=1, count=1)


def test_pipelined_step_matches_fused(make_sim, golden):
    pc.check_pipeline_identical(make_sim, golden, n=1, steps=1, settle=1, pipelines=(0, 1), first_state=9)


def test_row_pass_of_the_narrowphase_matches_fused(make_sim, golden, monkeypatch):
    """the k_narrow instance of large batches (row pass: four light pairs per wavefront), forced on for one env"""
    monkeypatch.setenv("SO101_NARROW_ROWS", "1")
    pc.check_pipeline_identical(make_sim, golden, n=1, steps=1, settle=1, pipelines=(0, 1), first_state=9)


@SLOW
def test_chained_and_merged_steps_match_fused(make_sim, golden):
    # two envs on concurrently alive emulated wavefronts: every queue hand-off of so101_chain.hpp (pipeline 2: persistent k_chain)
    # against the fused step, bit for bit.  (Pipeline 3 - narrowphase chunks inside the solve launches, the same queue code - is
    # compared on the GPU only, tests/test_gpu_parity.py: the emulated run costs a minute of the CPU suite per pipeline.)
    pc.check_pipeline_identical(make_sim, golden, n=2, steps=1, settle=2, pipelines=(0, 2))


def test_pgs_forward_matches_oracle_pgs(make_sim, blobs):
    pc.check_pgs_forward(make_sim, blobs, n=1, iterations=30)
