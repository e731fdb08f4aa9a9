"""
Command line of the MI355X build: the ``glimpse`` / ``fit`` / ``stats`` / ``log`` commands of ``tapqir``
(tapqir/main.py:66-318, 321-576, 873-884, 1387-1488) over the same workspace (``<cd>/.tapqir/config.yaml``, ``loginfo``, ``<model>_model.tpqr``,
``<model>_params.tpqr``, ``<model>_summary.csv``).  Options, defaults and exit codes (0 / 1) follow the reference;
``--cpu`` exits with 1 because the SVI step has no CPU path here.  ``glimpse`` takes its inputs from flags or from
``config.yaml`` (there are no interactive prompts).  Plotting (``show``) and the kinetics commands are outside the
hot-path scope of this build.

    python -m tapqir_amd --cd <dir> fit --model cosmos --cuda --num-iter 0 --no-input
"""

import logging
import sys
from enum import Enum
from pathlib import Path
from typing import List, Optional

import typer
import yaml

from tapqir_amd import __version__

app = typer.Typer()

PRIOR_DEFAULTS = {  # main.py:1431-1439
    "background_mean_std": 1000, "background_std_std": 100, "lamda_rate": 1, "height_std": 10000,
    "width_min": 0.75, "width_max": 2.25, "proximity_rate": 1, "gain_std": 50,
}
CONFIG_DEFAULTS = {  # main.py:1423-1445
    "P": 14, "nbatch-size": 10, "fbatch-size": 512, "learning-rate": 0.005, "num-channels": 1, "cuda": True,
    "matlab": False, "priors": dict(PRIOR_DEFAULTS), "offset-x": 10, "offset-y": 10, "offset-P": 30, "bin-size": 1,
}
DEFAULTS = {"cd": Path.cwd()}


class avail_models(str, Enum):  # main.py:27-28: the keys of tapqir.models.models
    cosmos = "cosmos"
    crosstalk = "crosstalk"
    hmm = "cosmos+hmm"


def _default(key):
    return lambda: DEFAULTS.get(key, CONFIG_DEFAULTS.get(key))


def _version(value: bool):
    if value:
        typer.echo(f"Tapqir-AMD version: {__version__}")
        raise typer.Exit()


def _write_config(cd: Path):
    with open(cd / ".tapqir" / "config.yaml", "w") as f:
        yaml.dump({k: v for k, v in DEFAULTS.items() if k != "cd"}, f, sort_keys=False)


def _build_model(name: str, logger, **settings):
    from tapqir_amd.exceptions import HipExtensionError
    from tapqir_amd.models import models

    try:
        return models[name](**settings)
    except (NotImplementedError, HipExtensionError):
        logger.exception(f"Model {name} is not available in this build")
        return None


@app.command()
def glimpse(
    dataset: str = typer.Option(_default("dataset"), help="Dataset name"),
    P: int = typer.Option(_default("P"), "--aoi-size", "-P", min=5, max=50, help="AOI image size - number of pixels along the axis"),
    offset_x: int = typer.Option(_default("offset-x"), "--offset-x", min=0, help="x-axis position of the top-left corner of the offset region"),
    offset_y: int = typer.Option(_default("offset-y"), "--offset-y", min=0, help="y-axis position of the top-left corner of the offset region"),
    offset_P: int = typer.Option(_default("offset-P"), "--offset-P", min=5, help="Offset region size - number of pixels along the axis"),
    bin_size: int = typer.Option(_default("bin-size"), "--bin-size", min=1, max=21, help="Offset histogram bin size (odd number)"),
    frame_range: bool = typer.Option(_default("frame-range"), help="Specify frame range."),
    frame_start: Optional[int] = typer.Option(_default("frame-start"), min=0, help="Starting frame."),
    frame_end: Optional[int] = typer.Option(_default("frame-end"), min=1, help="Ending frame."),
    use_offtarget: bool = typer.Option(_default("use-offtarget"), help="Use off-target AOI locations."),
    num_channels: int = typer.Option(_default("num-channels"), "--num-channels", "-C", min=1, help="Number of color channels"),
    name: Optional[List[str]] = typer.Option(None, help="Channel name (once per channel)"),
    glimpse_folder: Optional[List[Path]] = typer.Option(None, exists=True, file_okay=False, resolve_path=True),
    driftlist: Optional[List[Path]] = typer.Option(None, exists=True, dir_okay=False, resolve_path=True),
    ontarget_aoiinfo: Optional[List[Path]] = typer.Option(None, exists=True, dir_okay=False, resolve_path=True),
    offtarget_aoiinfo: Optional[List[Path]] = typer.Option(None, exists=True, dir_okay=False, resolve_path=True),
    ontarget_labels: Optional[List[Path]] = typer.Option(None, exists=True, dir_okay=False, resolve_path=True),
    offtarget_labels: Optional[List[Path]] = typer.Option(None, exists=True, dir_okay=False, resolve_path=True),
    overwrite: bool = typer.Option(True, "--overwrite", "-w", help="Overwrite defaults values."),
    no_input: bool = typer.Option(False, "--no-input", help="Accepted for compatibility (there are no prompts)."),
    labels: bool = typer.Option(False, "--labels", "-l", help="Add on-target binding labels."),
    progress_bar=None,
):
    """
    Extract AOIs from raw glimpse images (tapqir/main.py:66-318) into ``<cd>/data.tpqr``.

    Needs, per colour channel: the header/glimpse folder, the aoiinfo file of the target molecules (on-target AOIs),
    optionally the aoiinfo file of the off-target control locations, and the driftlist file.  Values not given as flags
    are taken from ``.tapqir/config.yaml`` (``channels: [{name, glimpse-folder, driftlist, ontarget-aoiinfo, ...}]``).
    """
    from tapqir_amd.exceptions import HipExtensionError
    from tapqir_amd.imscroll import read_glimpse

    cd = DEFAULTS["cd"]
    logger = logging.getLogger("tapqir")
    DEFAULTS.update({"dataset": dataset, "P": P, "offset-P": offset_P, "offset-x": offset_x, "offset-y": offset_y,
                     "bin-size": bin_size, "frame-range": bool(frame_range), "frame-start": frame_start,
                     "frame-end": frame_end, "use-offtarget": bool(use_offtarget), "num-channels": num_channels,
                     "labels": labels})
    flags = {"name": name, "glimpse-folder": glimpse_folder, "driftlist": driftlist, "ontarget-aoiinfo": ontarget_aoiinfo,
             "offtarget-aoiinfo": offtarget_aoiinfo, "ontarget-labels": ontarget_labels, "offtarget-labels": offtarget_labels}
    channels = [dict(ch) for ch in DEFAULTS.get("channels") or []]
    for c in range(num_channels):
        if len(channels) < c + 1:
            channels.append({})
        for key, values in flags.items():
            if values and c < len(values):
                channels[c][key] = str(values[c])
        needed = ["name", "glimpse-folder", "driftlist", "ontarget-aoiinfo"] + (["offtarget-aoiinfo"] if use_offtarget else [])
        missing = [key for key in needed if channels[c].get(key) is None]
        if missing:
            logger.error(f"Channel #{c}: missing {', '.join('--' + k for k in missing)} (flag or config.yaml)")
            raise typer.Exit(1)
        if labels:
            channels[c].setdefault("ontarget-labels", None)
            channels[c].setdefault("offtarget-labels", None)
    DEFAULTS["channels"] = channels[:num_channels]
    if overwrite:
        _write_config(cd)

    logger.info("Extracting AOIs ...")
    try:
        read_glimpse(path=cd, progress_bar=progress_bar, **{k: v for k, v in DEFAULTS.items() if k != "cd"})
    except HipExtensionError:
        logger.exception("Failed to extract AOIs: the extraction needs an AMD GPU and the built HIP library")
        raise typer.Exit(1)
    logger.info("Extracting AOIs: Done")


@app.command()
def fit(
    model: avail_models = typer.Option("cosmos", help="Tapqir model"),
    S: int = typer.Option(1, "--num-states", "-S", help="Number of spot states"),
    cuda: bool = typer.Option(_default("cuda"), "--cuda/--cpu", help="Run computations on GPU or CPU", show_default=False),
    nbatch_size: int = typer.Option(_default("nbatch-size"), "--nbatch-size", "-nbs", help="AOI batch size"),
    fbatch_size: int = typer.Option(_default("fbatch-size"), "--fbatch-size", "-fbs", help="Frame batch size"),
    learning_rate: float = typer.Option(_default("learning-rate"), "--learning-rate", "-lr", help="Learning rate"),
    num_iter: int = typer.Option(0, "--num-iter", "-it", help="Number of iterations (0 = until converged)"),
    k_max: int = typer.Option(2, "--k-max", "-k", help="Maximum number of spots per image"),
    matlab: bool = typer.Option(_default("matlab"), "--matlab", help="Save parameters in matlab format"),
    funsor: bool = typer.Option(False, "--funsor/--pyro", help="Accepted for compatibility; ignored"),
    pykeops: bool = typer.Option(True, "--pykeops/--no-pykeops", help="Accepted for compatibility; ignored"),
    overwrite: bool = typer.Option(True, "--overwrite", "-w", help="Overwrite defaults values."),
    no_input: bool = typer.Option(False, "--no-input", help="Accepted for compatibility (there are no prompts)."),
    progress_bar=None,
):
    """
    Fit the data to the selected model (cosmos, crosstalk).
    """
    from tapqir_amd.exceptions import CudaOutOfMemoryError, HipExtensionError, TapqirFileNotFoundError

    cd = DEFAULTS["cd"]
    logger = logging.getLogger("tapqir")
    settings = {"S": S, "K": k_max, "device": "cuda" if cuda else "cpu", "dtype": "double", "use_pykeops": pykeops,
                "priors": {k: float(v) for k, v in DEFAULTS.get("priors", PRIOR_DEFAULTS).items()}}
    if overwrite:
        DEFAULTS.update({"cuda": cuda, "nbatch-size": nbatch_size, "fbatch-size": fbatch_size,
                         "learning-rate": learning_rate, "matlab": matlab})
        _write_config(cd)

    logger.info("Fitting the data ...")
    m = _build_model(model.value, logger, **settings)
    if m is None:
        raise typer.Exit(1)
    try:
        m.load(cd)
    except TapqirFileNotFoundError as err:
        logger.exception(f"Failed to load {err.name} file")
        raise typer.Exit(1)
    try:
        m.init(learning_rate, nbatch_size, fbatch_size)
        m.run(num_iter, progress_bar=progress_bar)
    except HipExtensionError:
        logger.exception("Failed to fit the data: the SVI step needs an AMD GPU (--cuda) and the built HIP library")
        raise typer.Exit(1)
    except CudaOutOfMemoryError:
        logger.exception("Failed to fit the data")
        raise typer.Exit(1)
    logger.info("Fitting the data: Done")

    logger.info("Computing stats ...")
    try:
        m.compute_stats(save_matlab=matlab)
    except CudaOutOfMemoryError:
        logger.exception("Failed to compute stats")
        raise typer.Exit(1)
    logger.info("Computing stats: Done")


@app.command()
def stats(
    model: avail_models = typer.Option("cosmos", help="Tapqir model"),
    cuda: bool = typer.Option(_default("cuda"), "--cuda/--cpu", help="Run computations on GPU or CPU", show_default=False),
    nbatch_size: int = typer.Option(_default("nbatch-size"), "--nbatch-size", "-nbs", help="AOI batch size"),
    fbatch_size: int = typer.Option(_default("fbatch-size"), "--fbatch-size", "-fbs", help="Frame batch size"),
    matlab: bool = typer.Option(_default("matlab"), "--matlab", help="Save parameters in matlab format"),
    funsor: bool = typer.Option(False, "--funsor/--pyro", help="Accepted for compatibility; ignored"),
    no_input: bool = typer.Option(False, "--no-input", help="Accepted for compatibility (there are no prompts)."),
):
    """
    Compute credible intervals and classification statistics from the last checkpoint.
    """
    from tapqir_amd.exceptions import CudaOutOfMemoryError, HipExtensionError, TapqirFileNotFoundError

    cd = DEFAULTS["cd"]
    logger = logging.getLogger("tapqir")
    logger.info("Computing stats ...")
    m = _build_model(model.value, logger, device="cuda" if cuda else "cpu", dtype="double")
    if m is None:
        raise typer.Exit(1)
    try:
        m.load(cd)
    except TapqirFileNotFoundError:
        logger.exception("Failed to load data file")
        raise typer.Exit(1)
    try:
        m.load_checkpoint(param_only=True)
        m.nbatch_size = nbatch_size
        m.fbatch_size = fbatch_size
        m.compute_stats(save_matlab=matlab)
    except TapqirFileNotFoundError as err:
        logger.exception(f"Failed to load {err.name} file")
        raise typer.Exit(1)
    except HipExtensionError:
        logger.exception("Failed to compute stats: the posterior read-out needs an AMD GPU (--cuda) and the built HIP library")
        raise typer.Exit(1)
    except CudaOutOfMemoryError:
        logger.exception("Failed to compute stats")
        raise typer.Exit(1)
    logger.info("Computing stats: Done")


@app.command()
def log():
    """Show logging info (``.tapqir/loginfo``)."""
    path = DEFAULTS["cd"] / ".tapqir" / "loginfo"
    if path.is_file():
        typer.echo(path.read_text())


@app.callback()
def main(
    cd: Path = typer.Option(Path.cwd(), help="Change working directory.", show_default=False, exists=True,
                            file_okay=False, dir_okay=True),
    version: Optional[bool] = typer.Option(None, "--version", callback=_version, is_eager=True,
                                           help="Show version and exit."),
):
    """
    Bayesian analysis of co-localization single-molecule microscopy image data on AMD MI355X.

    Initializes a Tapqir workspace in the working directory: a ``.tapqir`` sub-directory with ``config.yaml``,
    ``loginfo`` and the files written by ``fit``.
    """
    DEFAULTS.clear()
    DEFAULTS["cd"] = cd
    tp = cd / ".tapqir"
    tp.mkdir(exist_ok=True)
    cfg = tp / "config.yaml"
    if not cfg.is_file():
        DEFAULTS.update({k: (dict(v) if isinstance(v, dict) else v) for k, v in CONFIG_DEFAULTS.items()})
        _write_config(cd)
        typer.echo(f"Initialized Tapqir at {tp}.")

    # the package's modules log under "tapqir_amd.*" (getLogger(__name__)); "tapqir" is kept for code written against the
    # reference's logger name (main.py:1353-1372)
    ch = logging.StreamHandler(sys.stdout)
    ch.setLevel(logging.INFO)
    ch.setFormatter(logging.Formatter("%(levelname)s - %(message)s"))
    fh = logging.FileHandler(tp / "loginfo")
    fh.setLevel(logging.DEBUG)
    fh.setFormatter(logging.Formatter(fmt="%(asctime)s - %(levelname)s - %(message)s", datefmt="%m/%d/%Y %I:%M %p"))
    for name in ("tapqir_amd", "tapqir"):
        lg = logging.getLogger(name)
        lg.setLevel(logging.DEBUG)
        for h in list(lg.handlers):  # repeated invocations in one process (tests) must not stack handlers
            lg.removeHandler(h)
            h.close()
        lg.addHandler(ch)
        lg.addHandler(fh)
    logger = logging.getLogger("tapqir_amd")

    with open(cfg) as f:
        DEFAULTS.update(yaml.safe_load(f) or {})
    logger.info(f"Configuration options are read from {cfg}.")


if __name__ == "__main__":
    app()
