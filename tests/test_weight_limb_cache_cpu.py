"""Host logic of dense.weight_limbs (the limb images of a step's weight operands, split once per optimizer step in one launch),
with the split launch and the library stubbed out: which images are (re)split when."""
import types

import pytest
import torch


@pytest.fixture
def cache(monkeypatch):
    from tf_gnn_samples_amd import _lib, dense as DN
    launches = []

    class Lib:
        @staticmethod
        def relgnn_limb_elements(r, c):
            return ((r + 31) // 32) * 32 * c * 3
    monkeypatch.setattr(DN, "_split_weight_images", lambda ims: launches.append(sum(len(i.items) for i in ims)))
    monkeypatch.setattr(_lib, "load_library", lambda: Lib())
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda d=None: types.SimpleNamespace(cuda_stream=0))
    monkeypatch.setattr(DN, "_WEIGHT_LIMBS", DN._PerStream(limit=8))
    from tf_gnn_samples_amd import config
    monkeypatch.setattr(config.settings, "weight_limb_cache", "1")
    return DN, launches


def _step(DN, layers, dense):
    for ks in layers:
        DN.weight_limbs(ks, DN.WEIGHT_NN)
    DN.weight_limbs(dense, DN.WEIGHT_NN)
    DN.weight_limbs(dense, DN.WEIGHT_NT)
    for ks in layers[::-1]:
        DN.weight_limbs(ks, DN.WEIGHT_NT)


def test_one_launch_per_step_after_the_first(cache):
    DN, launches = cache
    layers = [[torch.nn.Parameter(torch.randn(32, 32)) for _ in range(3)] for _ in range(3)]
    dense = torch.nn.Parameter(torch.randn(32, 32))
    _step(DN, layers, dense)
    assert launches == [3, 3, 3, 1, 1, 3, 3, 3]            # first step: every image on its own
    for _ in range(3):
        del launches[:]
        DN.weights_changed()
        _step(DN, layers, dense)
        assert launches == [20]                            # 8 images, 20 matrices, one launch
    del launches[:]
    _step(DN, layers, dense)                               # nothing changed (an evaluation pass): nothing is split
    assert launches == []


def test_in_place_writes_and_dead_tensors(cache):
    DN, launches = cache
    a, b = torch.nn.Parameter(torch.randn(32, 32)), torch.nn.Parameter(torch.randn(32, 32))
    DN.weight_limbs(a, DN.WEIGHT_NN); DN.weight_limbs(b, DN.WEIGHT_NN)
    del launches[:]
    with torch.no_grad():
        a.add_(1.0)                                        # the version counter moves: only a's image is stale
    DN.weight_limbs(b, DN.WEIGHT_NN)
    assert launches == []
    DN.weight_limbs(a, DN.WEIGHT_NN)
    assert launches == [1]
    v = a.view(32, 32)                                     # a view shares the parameter's identity and version
    DN.weight_limbs(v, DN.WEIGHT_NN)
    assert launches == [1]
    DN.weights_changed(); DN.weights_changed()             # two updates without a request in between
    del launches[:]
    DN.weight_limbs(a, DN.WEIGHT_NN)                       # b's image was not used during the last step: dropped, not re-split
    assert launches == [1]
    table = next(iter(DN._WEIGHT_LIMBS.values()))
    assert len(table) == 1
    del a, v
    DN.weights_changed()
    DN.weight_limbs(b, DN.WEIGHT_NN)                       # the dead parameter's image is forgotten
    assert len(table) == 1 and launches == [1, 1]
