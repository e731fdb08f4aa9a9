"""
Command line (tapqir_amd/main.py) against the reference's own CLI smoke test (test/test_tapqir.py:20-140): canonical
simulation parameters, N=2, F=5, C=1, P=14, one iteration, exit codes.  ``--cpu`` must fail loudly (exit 1): the SVI step
has no CPU path in this build.
"""

import pytest
import yaml
from typer.testing import CliRunner

from tapqir_amd.main import app
from tapqir_amd.utils.dataset import save
from tapqir_amd.utils.simulate import TEST_PARAMS, simulate

runner = CliRunner()


@pytest.fixture(params=["cosmos", "crosstalk"])
def dataset_path(request, tmp_path):
    params = dict(TEST_PARAMS)
    if request.param == "crosstalk":
        params["alpha"] = [[1.0]]  # test/test_tapqir.py:29
    save(simulate(2, 2, 5, 1, 14, params=params), tmp_path)
    return tmp_path


def fit_cmd(path, model, device):
    return ["--cd", str(path), "fit", "--model", model, "-S", "1", "--learning-rate", "0.005", "--nbatch-size", "2",
            "--fbatch-size", "5", "--num-iter", "1", device, "--no-input"]


def test_workspace_and_cpu_refusal(dataset_path):
    result = runner.invoke(app, fit_cmd(dataset_path, "cosmos", "--cpu"))
    assert result.exit_code == 1  # no CPU fallback
    assert "AMD GPU" in result.output
    cfg = yaml.safe_load(open(dataset_path / ".tapqir" / "config.yaml"))
    assert cfg["nbatch-size"] == 2 and cfg["fbatch-size"] == 5 and cfg["learning-rate"] == 0.005 and cfg["cuda"] is False
    assert cfg["priors"]["height_std"] == 10000 and cfg["offset-P"] == 30  # main.py:1423-1445
    assert (dataset_path / ".tapqir" / "loginfo").is_file()
    assert runner.invoke(app, ["--cd", str(dataset_path), "log"]).exit_code == 0


def test_missing_data_and_unavailable_model(tmp_path):
    assert runner.invoke(app, fit_cmd(tmp_path, "cosmos", "--cuda")).exit_code == 1  # no data.tpqr
    assert runner.invoke(app, fit_cmd(tmp_path, "cosmos+hmm", "--cuda")).exit_code == 1
    result = runner.invoke(app, ["--version"])
    assert result.exit_code == 0 and "version" in result.output


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["cosmos", "crosstalk"])
def test_commands_cuda(dataset_path, model):
    """test/test_tapqir.py:96-127 (fit) and 74-88 (stats, run there on the CPU)."""
    result = runner.invoke(app, fit_cmd(dataset_path, model, "--cuda"))
    assert result.exit_code == 0, result.output
    for f in (f".tapqir/{model}_model.tpqr", f"{model}_params.tpqr", f"{model}_summary.csv"):
        assert (dataset_path / f).is_file(), f
    result = runner.invoke(app, ["--cd", str(dataset_path), "stats", "--model", model, "--nbatch-size", "2",
                                 "--fbatch-size", "5", "--cuda", "--matlab", "--no-input"])
    assert result.exit_code == 0, result.output
    assert (dataset_path / f"{model}_params.mat").is_file()


def glimpse_cmd(path, cfg):
    ch = cfg["channels"][0]
    return ["--cd", str(path), "glimpse", "--dataset", "synthetic", "-P", str(cfg["P"]), "--offset-x", str(cfg["offset-x"]),
            "--offset-y", str(cfg["offset-y"]), "--offset-P", str(cfg["offset-P"]), "--bin-size", "1", "--use-offtarget",
            "-C", "1", "--name", ch["name"], "--glimpse-folder", ch["glimpse-folder"], "--driftlist", ch["driftlist"],
            "--ontarget-aoiinfo", ch["ontarget-aoiinfo"], "--offtarget-aoiinfo", ch["offtarget-aoiinfo"], "--no-input"]


def test_glimpse_command_needs_its_inputs_and_a_gpu(tmp_path):
    """tapqir/main.py:66-318: without the per-channel files the command fails (exit 1); with them the extraction itself
    needs the GPU (no CPU path) -- on a box without one that is exit 1 too, with the inputs recorded in config.yaml."""
    import torch
    from glimpse_fixture import write_experiment

    assert runner.invoke(app, ["--cd", str(tmp_path), "glimpse", "--dataset", "x", "--no-input"]).exit_code == 1
    cfg, _ = write_experiment(tmp_path / "raw", F=4, labels=False)
    result = runner.invoke(app, glimpse_cmd(tmp_path, cfg))
    saved = yaml.safe_load(open(tmp_path / ".tapqir" / "config.yaml"))
    assert saved["channels"][0]["name"] == "dye0" and saved["offset-P"] == 20 and saved["use-offtarget"] is True
    if not torch.cuda.is_available():
        assert result.exit_code == 1 and "AMD GPU" in result.output


@pytest.mark.gpu
def test_glimpse_then_fit(tmp_path):
    """Raw frames -> data.tpqr -> one SVI iteration, all through the command line."""
    import torch
    from glimpse_fixture import write_experiment
    from oracle import glimpse as og

    cfg, _ = write_experiment(tmp_path / "raw", F=6, labels=False)
    result = runner.invoke(app, glimpse_cmd(tmp_path, cfg))
    assert result.exit_code == 0, result.output
    saved = torch.load(tmp_path / "data.tpqr", weights_only=False)
    assert torch.equal(saved["images"], og.read_glimpse(**cfg)["images"])
    result = runner.invoke(app, fit_cmd(tmp_path, "cosmos", "--cuda"))
    assert result.exit_code == 0, result.output
    assert (tmp_path / "cosmos_summary.csv").is_file()
    # a second run picks the channel files up from config.yaml
    result = runner.invoke(app, ["--cd", str(tmp_path), "glimpse", "--no-input"])
    assert result.exit_code == 0, result.output
