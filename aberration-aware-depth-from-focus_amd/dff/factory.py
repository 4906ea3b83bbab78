"""dff/factory.py of the reference: build the train / test lens from the YAML dict (dff/factory.py:4-31) and pick a dataset
(:33-55).  `get_lens` is the reference's logic on this package's PSFNet / ThinLens.  The reference's datasets read
Matterport3D / FlyingThings3D / Middlebury files with cv2, skimage and torchvision (not part of the hot path, not installed
here): `get_dataset` offers the seeded synthetic RGB-D set the benchmarks use and names what is missing otherwise."""
import torch

from deeplens.psfnet import PSFNet, ThinLens
from aadff.synth import synth_depth_mm, synth_rgb


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
    sets = []
    for split in ("train", "test"):
        name = args[split]["dataset"]
        if name != "Synthetic":
            raise NotImplementedError(f"dataset '{name}' reads files through cv2 / skimage / torchvision in the reference "
                                      "(dff/dataset.py); only 'Synthetic' is provided here")
        sets.append(SyntheticRGBD(args[split].get("n", 64), args["res"], seed=1000 if split == "train" else 2000))
    return tuple(sets)
