"""CPU-only: which sets of work ranges a call of nbnxm_gpu_launch_kernel / nbnxm_gpu_launch_kernel_part launches and which call carries
the trailing workgroups (host arithmetic behind nbnxm_hip_query_launch_plan, include/nbnxm_hip.h).  The invariants are the ones a
default-on multi-rank path depends on (the two-part local launch of the domain step, csrc/halo_exchange.hip):
  * part 1 + part 2 together launch every set exactly once and the tail exactly once, like a plain launch (part 0) does;
  * a call that launches nothing carries no tail, i.e. consumes no state of the step — before commit c622397 part 2 on a list too
    short for two sets dropped a pending rolling-prune part and marked the spare force buffer as zeroed (stale forces next step);
  * no launch is empty and no set reaches beyond the range arrays, for odd and even range counts."""
import ctypes as C

import pytest

import fep_testlib as tl

pkg = tl.pkg


def plan(part, work_parts, num_ranges):
    lib = pkg.hip_lib()
    first, nsets, per_set, tail = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    lib.nbnxm_hip_query_launch_plan(C.c_int(part), C.c_int(work_parts), C.c_int(num_ranges), C.byref(first), C.byref(nsets), C.byref(per_set),
                                    C.byref(tail))
    return first.value, nsets.value, per_set.value, tail.value


RANGE_COUNTS = [0, 1, 2, 3, 7, 8, 1023, 4096, 5120, 8192, 8193, 10240]


@pytest.mark.parametrize("work_parts", [1, 2])
@pytest.mark.parametrize("num_ranges", RANGE_COUNTS)
def test_parts_cover_every_set_once_and_the_tail_once(work_parts, num_ranges):
    launched = {0: [], 1: [], 2: []}
    tails = {0: 0, 1: 0, 2: 0}
    for part in (0, 1, 2):
        first, nsets, per_set, tail = plan(part, work_parts, num_ranges)
        assert nsets in (0, 1, 2) and first in (0, 1)
        if nsets == 0:
            assert tail == 0, "a call that queues no kernel must not consume the step's state"
        else:
            assert per_set >= 1, "no empty launch"
            assert (first + nsets) * per_set <= num_ranges, "set beyond the range arrays"
        launched[part] = [(first + k, per_set) for k in range(nsets)]
        tails[part] = tail
    # the plain launch covers all ranges and carries the tail (unless the list has no ranges at all)
    assert sum(n for _, n in launched[0]) == num_ranges
    assert tails[0] == (1 if num_ranges > 0 else 0)
    # the two parts together are the plain launch
    assert launched[1] + launched[2] == launched[0]
    assert tails[1] + tails[2] == tails[0]
    # with two sets the tail rides with the second part (behind the non-local kernel), else with the first
    if launched[2]:
        assert tails[2] == 1 and tails[1] == 0
    elif num_ranges > 0:
        assert tails[1] == 1


def test_two_sets_need_an_even_partition_made_for_two_parts():
    assert plan(0, 2, 10240) == (0, 2, 5120, 1)
    assert plan(1, 2, 10240) == (0, 1, 5120, 0)
    assert plan(2, 2, 10240) == (1, 1, 5120, 1)
    # one set: the first part is the whole launch, the second part is nothing and takes nothing
    assert plan(1, 1, 5120) == (0, 1, 5120, 1)
    assert plan(2, 1, 5120) == (0, 0, 5120, 0)
    # (guard) an odd count is never split
    assert plan(1, 2, 4097) == (0, 1, 4097, 1)
    assert plan(2, 2, 4097) == (0, 0, 4097, 0)
