"""sample()'s return value (hamiltorch_amd/samplelist.py): a real ``list`` whose rows stay one tensor until something needs
the individual objects.  These tests pin that every way the reference's notebooks and tests use the returned list
(S:957-959, S:1084-1091: torch.stack / torch.cat / indexing / slicing / iteration / len / predict_model) gives exactly what the
plain list gives."""
import copy
import pickle

import numpy as np
import pytest
import torch

from hamiltorch_amd import samplelist as SL


def _pair(S=200, shape=(3, 2)):
    t = torch.arange(S * int(np.prod(shape)), dtype=torch.float32).reshape((S,) + shape)
    return t, SL.SampleList(t), list(t.unbind(0))


def test_is_a_list_with_the_rows_of_the_plain_list():
    t, x, ref = _pair()
    assert isinstance(x, list) and len(x) == len(ref) == 200 and not x._done
    for i in (0, 5, 199, -1, -200):
        assert torch.equal(x[i], ref[i])
    for bad in (200, -201):
        with pytest.raises(IndexError):
            x[bad]
    assert not x._done                                       # len and integer indexing never build the rows


def test_stack_and_cat_answer_from_the_backing_tensor():
    t, x, ref = _pair()
    s = torch.stack(x)
    assert torch.equal(s, torch.stack(ref)) and s.data_ptr() != t.data_ptr() and not x._done      # a copy, like torch.stack
    assert torch.equal(torch.stack(x, 0), torch.stack(ref, 0)) and torch.equal(torch.stack(x, dim=0), torch.stack(ref))
    assert torch.equal(torch.cat(x), torch.cat(ref)) and torch.equal(torch.cat(x, dim=0), torch.cat(ref, 0)) and not x._done
    s[0].zero_()
    assert torch.equal(x[0], ref[0])                         # the copy does not alias the samples
    # any other layout / function materialises and agrees
    assert torch.equal(torch.stack(x, dim=1), torch.stack(ref, dim=1)) and x._done
    _, x, _ = _pair()
    assert torch.equal(torch.cat(x, dim=1), torch.cat(ref, dim=1)) and x._done
    _, x, _ = _pair()
    assert torch.equal(torch.vstack(x), torch.vstack(ref)) and torch.equal(torch.hstack(x), torch.hstack(ref))
    _, x, _ = _pair()
    out = torch.empty(200, 3, 2)
    torch.stack(x, out=out)
    assert torch.equal(out, t)
    # one chain: rows are (D,)
    t1 = torch.arange(400.0).reshape(200, 2)
    x1 = SL.SampleList(t1)
    assert x1[3].shape == (2,) and torch.equal(torch.stack(x1), t1) and torch.equal(torch.cat(x1), t1.reshape(-1)) and not x1._done
    # the reference's notebook idiom: coords = torch.cat(params).reshape(len(params), -1)
    assert torch.equal(torch.cat(x1).reshape(len(x1), -1), t1)


def test_python_protocols_match_the_plain_list():
    t, _, ref = _pair()
    mk = lambda: SL.SampleList(t)                                                                     # noqa: E731
    x = mk(); sl = x[10:20]
    assert type(sl) is list and len(sl) == 10 and torch.equal(sl[0], ref[10]) and x._done
    x = mk(); assert type(x[::50]) is list and len(x[::50]) == 4
    x = mk(); assert sum(1 for _ in x) == 200 and x._done
    for conv in (list, tuple):
        x = mk(); c = conv(x)
        assert len(c) == 200 and torch.equal(c[7], ref[7]) and type(c) is conv
    x = mk(); assert torch.equal(list(reversed(x))[0], ref[-1])
    x = mk(); y = x + [torch.zeros(3, 2)]; assert type(y) is list and len(y) == 201 and torch.equal(y[5], ref[5])
    x = mk(); x.append(torch.zeros(3, 2)); assert len(x) == 201 and torch.equal(x[200], torch.zeros(3, 2)) and torch.equal(x[1], ref[1])
    x = mk(); x.extend([torch.ones(3, 2)] * 2); assert len(x) == 202
    x = mk(); last = x.pop(); assert torch.equal(last, ref[-1]) and len(x) == 199
    x = mk(); x[3] = torch.zeros(3, 2); assert torch.equal(x[3], torch.zeros(3, 2)) and torch.equal(x[4], ref[4])
    x = mk(); del x[0]; assert len(x) == 199 and torch.equal(x[0], ref[1])
    x = mk(); assert len(x * 2) == 400
    x = mk(); assert "tensor" in repr(x) and "_Row" not in repr(x) and "_Row" not in str(mk())
    x = mk(); p = pickle.loads(pickle.dumps(x)); assert type(p) is list and torch.equal(p[3], ref[3])
    x = mk(); d = copy.deepcopy(x); assert len(d) == 200 and torch.equal(d[9], ref[9]) and d[9].data_ptr() != x[9].data_ptr()
    x = mk(); c = copy.copy(x); assert type(c) is list and len(c) == 200
    x = mk(); assert [int(r.sum()) for r in x][:2] == [int(ref[0].sum()), int(ref[1].sum())]
    x = mk(); assert len([*x]) == 200
    x = mk(); assert torch.equal(list(enumerate(x))[199][1], ref[199]) and len(list(zip(mk(), range(3)))) == 3
    x = mk(); assert torch.equal(sorted(x, key=lambda r: -float(r.sum()))[0], ref[199])
    x = mk(); assert np.stack([r.numpy() for r in x]).shape == (200, 3, 2)
    x = mk(); a, b = x[:100], x[100:]; assert torch.equal(torch.stack(a + b), t)
    x = mk(); x.clear(); assert len(x) == 0 and x == []
    with pytest.raises(TypeError):
        hash(mk())
    # after materialising it simply is the list
    x = mk(); list(x)
    assert x._done and x._t is None and all(torch.is_tensor(r) for r in list.__iter__(x)) and torch.equal(torch.stack(x), t)
    assert torch.equal(SL.as_tensor(x), t) and torch.equal(SL.as_tensor(mk()), t) and torch.equal(SL.as_tensor(ref), t)


def test_rows_of_picks_the_plain_list_for_short_or_host_results():
    t = torch.zeros(10, 4, 3)
    assert type(SL.rows_of(t, False)) is list and type(SL.rows_of(torch.zeros(500, 4, 3), False)) is list      # host tensors: eager
    r = SL.rows_of(torch.zeros(500, 1, 3), True)
    assert type(r) is list and r[0].shape == (3,)


def test_predict_model_reads_the_lazy_list_without_materialising():
    import torch.nn as nn
    import hamiltorch_amd as ht
    from hamiltorch_amd import bnn
    torch.manual_seed(0)
    net = nn.Sequential(nn.Linear(2, 4), nn.Tanh(), nn.Linear(4, 1))
    D = sum(p.numel() for p in net.parameters())
    S = 80
    t = 0.3 * torch.randn(S, D)
    X, Y = torch.randn(12, 2), torch.randn(12, 1)
    lazy, plain = SL.SampleList(t), list(t.unbind(0))
    p1, l1 = ht.predict_model(net, lazy, x=X, y=Y, model_loss="regression", tau_out=2.0)
    assert not lazy._done
    p2, l2 = ht.predict_model(net, plain, x=X, y=Y, model_loss="regression", tau_out=2.0)
    assert torch.equal(p1, p2) and all(torch.equal(a, b) for a, b in zip(l1, l2))
    assert bnn.predict_stats["batched"] >= 2


def test_plain_list_plus_samplelist_holds_real_rows():
    """`[params_init] + samples` went through CPython's list_concat, which copies a list SUBCLASS's raw slots -
    the placeholders.  `__radd__` (tried first for a subclass) materialises; and a placeholder that does reach torch outside
    its list raises instead of dispatching to itself for ever."""
    from hamiltorch_amd.samplelist import SampleList, _ROW
    t = torch.randn(70, 3)
    first = torch.zeros(3)
    both = [first] + SampleList(t)
    assert type(both) is list and len(both) == 71 and all(isinstance(r, torch.Tensor) for r in both)
    assert torch.equal(torch.stack(both)[1:], t)
    s = SampleList(t)
    burn = [torch.ones(3)] * 2
    burn += s                                               # list_inplace_concat -> iteration
    assert len(burn) == 72 and torch.equal(torch.stack(burn[2:]), t)
    assert torch.equal(torch.stack(SampleList(t) + [first])[:-1], t)
    with pytest.raises(TypeError, match="placeholder"):
        torch.stack([_ROW, _ROW])
