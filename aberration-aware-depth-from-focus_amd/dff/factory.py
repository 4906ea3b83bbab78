"""dff/factory.py of the reference: build the train / test lens from the YAML dict (dff/factory.py:4-31) and pick a dataset
(:33-55).  `get_lens` is the reference's logic on this package's PSFNet / ThinLens.  `get_dataset` keeps the reference's
names and YAML keys (Matterport3D / FlyingThings3D for training, Middlebury2014 / 2021 / RealWorld for testing; files are read
with PIL and dff/exr.py) and adds the seeded synthetic RGB-D set the benchmarks use ('Synthetic')."""
import torch

from deeplens.psfnet import PSFNet, ThinLens
from aadff.synth import synth_depth_mm, synth_rgb
from .dataset import FlyingThings3D, Matterport3D, Middlebury, RealWorld


def _lens(spec, ks, sensor_res, device):
    name = spec["lens"]
    if name == "thinlens":
        lens = ThinLens(foc_len=spec["foc_len"], fnum=spec["fnum"], kernel_size=ks, sensor_size=[float(i) for i in spec["sensor_size"]],
                        sensor_res=sensor_res)
        return lens.to(device)
    lens = PSFNet(filename=name, sensor_res=sensor_res, kernel_size=ks, device=device)
    if spec.get("psfnet_path"):
        lens.load_net(spec["psfnet_path"])
    return lens


def get_lens(args):
    ks, sensor_res, device = args["ks"], args["res"], args["device"]
    return _lens(args["train"], ks, sensor_res, device), _lens(args["test"], ks, sensor_res, device)


class SyntheticRGBD(torch.utils.data.Dataset):
    """(all-in-focus RGB [3,H,W] in [0,1], depth [1,H,W] in metres) pairs from aadff.synth (seeded, no files)."""

    def __init__(self, n, resize, seed=0):
        self.n, self.res, self.seed = n, tuple(resize), seed

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        h, w = self.res
        return (torch.from_numpy(synth_rgb(h, w, seed=self.seed + 2 * idx)),
                torch.from_numpy(synth_depth_mm(h, w, seed=self.seed + 2 * idx + 1))[None] / 1e3)


def get_dataset(args):
    """(train_set, test_set) as dff/factory.py:33-55 selects them."""
    name = args["train"]["dataset"]
    if name == "Matterport3D":
        train_set = Matterport3D(args["train_aif_dir"], args["train_depth_dir"], resize=args["res"])
    elif name == "FlyingThings3D":
        train_set = FlyingThings3D(args["FlyingThings3D_train"], resize=args["res"])
    elif name == "Synthetic":
        train_set = SyntheticRGBD(args["train"].get("n", 64), args["res"], seed=1000)
    else:
        raise NotImplementedError(f"train dataset '{name}': 'Matterport3D', 'FlyingThings3D' and 'Synthetic' are provided")
    name = args["test"]["dataset"]
    if name in ("Middlebury2014", "Middlebury2021"):
        test_set = Middlebury(args[f"{name}_val"], resize=args["res"], train=False)
    elif name == "RealWorld":
        test_set = RealWorld(args["RealWorld_val"], resize=args["res"], depth=False)
    elif name == "Synthetic":
        test_set = SyntheticRGBD(args["test"].get("n", 64), args["res"], seed=2000)
    else:
        raise NotImplementedError(f"test dataset '{name}': 'Middlebury2014', 'Middlebury2021', 'RealWorld' and 'Synthetic' are provided")
    return train_set, test_set
