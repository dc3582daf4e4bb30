"""The config-JSON surface of the recipes (SURVEY 5 "Config" row): the committed fixtures carry the keys
and values of egs/wsj0-2mix/deep_clustering/config.json:1-27 and chimera/psa/config.json; run.py builds everything from
them (run.py:19-28)."""
import json
import os

import pytest
import torch

from onssen_amd import nn as onn
from onssen_amd.utils import AttrDict, build_optimizer

HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    with open(os.path.join(HERE, "golden", name)) as f:
        return AttrDict(json.load(f))


def test_attrdict_shim_behaves_like_the_recipes_need():
    args = load("config_dc.json")
    assert args.model_name == "dc" and args["model_options"]["hidden_dim"] == 600
    assert args.feature_options.batch_size == 16 and args.feature_options["hop_size"] == 64      # nested attribute access
    args.model = "anything"                                            # run.py:23 assigns attributes
    assert args["model"] == "anything"
    with pytest.raises(AttributeError):
        args.no_such_key
    assert dict(**args["model_options"]) == {"input_dim": 129, "hidden_dim": 600, "embedding_dim": 20, "num_layers": 3}


@pytest.mark.parametrize("cfg,cls,n_params", [("config_dc.json", "deep_clustering", 23_908_980), ("config_chimera_psa.json", "chimera", 32_866_038)])
def test_modules_build_from_the_config_fixture(cfg, cls, n_params):
    """``nn.<model>(**args['model_options'])`` with the shipped option keys; parameter counts of SURVEY 8a (A4, A9)."""
    args = load(cfg)
    model = getattr(onn, cls)(**args["model_options"])
    assert sum(p.numel() for p in model.parameters()) == n_params
    opt = build_optimizer(model.parameters(), args.optimizer_options)
    assert isinstance(opt, torch.optim.Adam) and opt.defaults["lr"] == 0.001
    assert torch.device(args.device).type == "cuda"                    # "cuda:0" / "cuda": PyTorch's name on ROCm too


@pytest.mark.gpu
def test_recipe_flow_from_config_on_the_gpu(monkeypatch):
    """run.py's sequence on the GPU box: config -> model -> loaders -> optimizer -> one training step -> tester.eval()
    on the evaluation loader (whole utterances, batch 1, label [Re, Im, sig_ref]).  The recipe's data_path is the author's
    corpus directory: without ONSSEN_SYNTHETIC_DATA=1 the factory refuses it (checked first)."""
    from onssen_amd.data import wsj0_2mix_dataloader as _dl
    monkeypatch.delenv("ONSSEN_SYNTHETIC_DATA", raising=False)
    with pytest.raises(FileNotFoundError):
        _dl("dc", load("config_dc.json").feature_options, "tr", "cuda:0")
    monkeypatch.setenv("ONSSEN_SYNTHETIC_DATA", "1")
    from onssen_amd import dist as odist
    from onssen_amd.data import wsj0_2mix_dataloader
    from onssen_amd.evaluate import tester_chimera, tester_dc
    from onssen_amd.loss import loss_chimera_psa, loss_dc
    for cfg, cls, loss_fn, tester_cls in (("config_dc.json", "deep_clustering", loss_dc, tester_dc),
                                          ("config_chimera_psa.json", "chimera", loss_chimera_psa, tester_chimera)):
        args = load(cfg)
        device = torch.device(args.device)
        args.model = getattr(onn, cls)(**args["model_options"])
        args.model.to(device)
        args.train_loader = wsj0_2mix_dataloader(args.model_name, args.feature_options, "tr", device)
        args.test_loader = wsj0_2mix_dataloader(args.model_name, args.feature_options, "tt", device)
        args.optimizer = build_optimizer(args.model.parameters(), args.optimizer_options)
        inp, lab = next(iter(args.train_loader))
        assert inp[0].shape == (args.feature_options.batch_size, 400, 129) and lab[0].dtype == torch.float64
        args.model.train()
        loss = odist.train_step(args.model, args.optimizer, loss_fn, inp, lab)
        assert loss == loss and abs(loss) < 1e9
        inp, lab = next(iter(args.test_loader))
        assert inp[0].shape[0] == 1 and len(lab) == 3 and lab[2].shape[:2] == (1, 2) and lab[2].shape[2] % 32 == 0
        assert lab[0].shape == inp[0].shape == lab[1].shape
        args.checkpoint_path = None
        sdr = tester_cls(args).eval()
        assert sdr == sdr and -60.0 < sdr < 60.0                        # random weights: a finite, unremarkable SI-SDR


# ---------------------------------------------------------------- optional hip_options keys (round 5; onssen_amd/options.py)
@pytest.fixture
def clean_options(monkeypatch):
    from onssen_amd import options
    for env, *_ in options.TABLE.values():
        monkeypatch.delenv(env, raising=False)
    saved = dict(options._configured)
    options._configured.clear()
    yield options
    options._configured.clear()
    options._configured.update(saved)


def test_reference_configs_load_unchanged_and_set_nothing(clean_options):
    """The reference's own config files carry none of the optional keys: they load as before and leave every switch at its default."""
    for cfg in ("config_dc.json", "config_chimera_psa.json"):
        args = load(cfg)
        kwargs = clean_options.apply_config(args)
        assert kwargs == dict(args["model_options"])
    assert all(v["source"] == "default" for v in clean_options.describe().values())
    assert clean_options.get("precision") == "bf16x3" and clean_options.get("recurrence") == "1"


def test_optional_keys_in_the_config_select_the_switches(clean_options, monkeypatch):
    args = load("config_dc.json")
    args["hip_options"] = {"precision": "f32", "recurrence": "steps", "dc_cluster": "steps", "fused_adam": False}
    args["model_options"]["fuse_first_layer"] = "auto"                 # the same keys may sit in model_options ...
    kwargs = clean_options.apply_config(args)
    assert "fuse_first_layer" not in kwargs and kwargs["hidden_dim"] == 600        # ... and are stripped for the constructor
    assert clean_options.get("precision") == "f32" and clean_options.get("recurrence") == "0"
    assert clean_options.get("dc_cluster") == "0" and clean_options.get("fused_adam") == "0"
    from onssen_amd.nn._core import precision
    assert precision() == "f32"
    # the constructors accept them too: nn.deep_clustering(**args['model_options']) keeps working with the extra keys
    m = onn.deep_clustering(129, 32, 2, 20, precision="bf16x3")
    assert clean_options.get("precision") == "bf16x3" and m.hidden_dim == 32
    with pytest.raises(TypeError, match="unexpected keyword argument 'hiden_dim'"):
        onn.deep_clustering(129, hiden_dim=32)
    with pytest.raises(ValueError, match="precision"):
        clean_options.configure(precision="fp8")
    # an environment variable that is set wins over the config (the operator's override for one run)
    monkeypatch.setenv("ONSSEN_PRECISION", "f32")
    assert clean_options.get("precision") == "f32" and clean_options.describe()["precision"]["source"] == "env"


def test_stated_world_size_must_match_the_launcher(clean_options, monkeypatch):
    args = load("config_dc.json")
    args["hip_options"] = {"world_size": 8}
    with pytest.raises(RuntimeError, match="world_size = 8"):
        clean_options.apply_config(args)
    monkeypatch.setenv("WORLD_SIZE", "8")
    clean_options.apply_config(args)


def test_every_onssen_switch_read_by_the_package_is_in_the_table():
    """No stray ``os.environ.get("ONSSEN_...")`` in the product path: one table, one resolution order."""
    import re
    root = os.path.join(os.path.dirname(HERE), "onssen_amd")
    stray = []
    for d, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py") and f != "options.py":
                for m in re.finditer(r'environ(?:\.get)?[\[(]\s*"(ONSSEN_[A-Z0-9_]+)"', open(os.path.join(d, f)).read()):
                    if m.group(1) not in ("ONSSEN_HIP_LIB",):             # where the library lives: read before anything else exists
                        stray.append((f, m.group(1)))
    assert not stray, stray


def test_packed_images_are_dropped_when_weights_may_have_moved():
    """A fused optimizer moves parameters without bumping ``_version`` (measured on the GPU box; the bug of rounds 3-4): the packed
    weight images must not rely on it.  ``build_optimizer``'s step hook and a train()/eval() switch bump the global epoch that is
    part of every image's key."""
    from onssen_amd.nn import _core
    model = onn.deep_clustering(129, 8, 1, 4)
    opt = build_optimizer(model.parameters(), {"name": "adam", "lr": 1e-3})
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    e0 = _core._WEIGHT_EPOCH[0]
    opt.step()
    assert _core._WEIGHT_EPOCH[0] == e0 + 1
    model.eval()
    assert _core._WEIGHT_EPOCH[0] == e0 + 2
    model.eval()                                   # no change of mode: nothing dropped
    assert _core._WEIGHT_EPOCH[0] == e0 + 2
    model.train()
    assert _core._WEIGHT_EPOCH[0] == e0 + 3


def test_cpu_tensors_raise_without_the_scaffolding_switch(monkeypatch):
    """No CPU fallback: an inference forward on a CPU tensor raises (always did), and so does a TRAINING forward unless the test
    scaffolding switch ONSSEN_CPU_AUTOGRAD=1 is set (tests/conftest.py sets it for the gloo tests)."""
    import pytest
    import torch
    from onssen_amd import nn as onn
    m = onn.deep_clustering(129, 8, 1, 20, dropout=0.0)
    x = torch.randn(2, 5, 129)
    monkeypatch.setenv("ONSSEN_CPU_AUTOGRAD", "0")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.train()([x])
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            m.eval()([x])
    monkeypatch.setenv("ONSSEN_CPU_AUTOGRAD", "1")
    assert m.train()([x])[0].shape == (2, 5, 129, 20)


def test_pipelines_refuse_a_cpu_model_loudly():
    """The stream pipelines (separation.DCPipeline / DCRaggedPipeline) exist on the device only: a CPU model raises before anything is
    allocated, and says what to use instead; tester_dc.eval checks the same predicate before it routes batches through the pipeline."""
    import pytest
    from onssen_amd import nn as onn
    from onssen_amd.separation import DCPipeline, DCRaggedPipeline
    m2 = onn.deep_clustering(129, 8, 2, 20).eval()
    assert "ROCm device" in DCRaggedPipeline.why_not(m2, 4)
    assert "num_layers = 2" in DCRaggedPipeline.why_not(onn.deep_clustering(129, 8, 3, 20).eval(), 4)
    with pytest.raises(RuntimeError, match="separate_dc"):
        DCRaggedPipeline(m2, 4, 64 * 20)
    with pytest.raises(RuntimeError, match="separate_dc"):
        DCPipeline(m2, 4, 64 * 20)
