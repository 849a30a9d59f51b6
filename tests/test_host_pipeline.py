"""CPU: the stream-selection rule of the throughput pipeline (vision3d_amd/detector/graph.py:choose_streams) on synthetic
timing models -- the rule itself has no device in it; PipelinedSecond.tune feeds it measured times
(tests/test_gpu_proposal.py runs that)."""
from vision3d_amd.detector.graph import choose_streams


def pipe_model(n_pipes, frame=550e-6, floor=300e-6, calls=None):
    """Streams whose hardware queues share a command-processor pipe do not overlap: time per frame = the frame time
    divided by the number of DISTINCT pipes in use, never below a floor (the part of a frame that needs the whole chip)."""
    def time_of(ids):
        if calls is not None:
            calls.append(tuple(ids))
        return max(frame / len({i % n_pipes for i in ids}), floor)
    return time_of


def test_picks_streams_on_distinct_pipes_and_stops_when_it_no_longer_pays():
    calls = []
    chosen, log = choose_streams(pipe_model(4, calls=calls), n_candidates=8, max_depth=4)
    assert len({i % 4 for i in chosen}) == len(chosen), chosen   # no two on one pipe
    assert len(chosen) == 2 and log[2] == 300e-6                  # 550/2 is under the floor: a third stream gains nothing
    assert set(log) == {2, 3} and log[3] >= 0.97 * log[2]
    assert len([c for c in calls if len(c) == 2]) == 28           # every pair was timed


def test_goes_deeper_while_it_pays_and_respects_the_maximum():
    chosen, log = choose_streams(pipe_model(8, frame=1200e-6, floor=100e-6), n_candidates=8, max_depth=4)
    assert len(chosen) == 4 and len(set(chosen)) == 4
    assert log[2] > log[3] > log[4]
    chosen, _ = choose_streams(pipe_model(8, frame=1200e-6, floor=100e-6), n_candidates=8, max_depth=3)
    assert len(chosen) == 3


def test_unlucky_creation_order_is_avoided():
    """Candidates 0 and 1 share a pipe (what taking streams in creation order would use): the best pair is another one."""
    def time_of(ids):
        pipes = {0: 0, 1: 0, 2: 1, 3: 0, 4: 2, 5: 1, 6: 0, 7: 3}
        return 500e-6 / len({pipes[i] for i in ids})
    chosen, log = choose_streams(time_of, 8, 2)
    assert len(chosen) == 2 and chosen != [0, 1] and log[2] == 250e-6


def test_degenerate_sizes():
    assert choose_streams(lambda ids: 1.0, 8, 1) == ([0], {})
    assert choose_streams(lambda ids: 1.0, 1, 4) == ([0], {})
    chosen, log = choose_streams(lambda ids: 1.0 / len(ids), 2, 4)
    assert chosen == [0, 1] and set(log) == {2}


def test_weight_change_epoch_is_one_integer_the_graph_runners_compare():
    """Second.notify_weights_changed (round 6): bumped by load_state_dict and by every train() / eval() switch, so that a
    captured-graph runner compares ONE integer per launch instead of ~120 parameter version counters."""
    import torch
    from vision3d_amd.core.config import second_car_cfg
    from vision3d_amd.detector import Second
    m = Second(second_car_cfg())
    e0 = m._weights_epoch
    m.eval()
    e1 = m._weights_epoch
    m.load_state_dict(m.state_dict())
    e2 = m._weights_epoch
    m.notify_weights_changed()
    assert e0 < e1 < e2 < m._weights_epoch
    with torch.no_grad():
        m.head.conv_cls.bias.add_(1.0)  # an in-place edit alone is NOT noticed by the counter (documented): the caller notifies
    assert m._weights_epoch == e2 + 1
