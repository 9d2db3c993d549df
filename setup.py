"""Packaging of the MI355X build: `pip install -e .` gives the `deepsignal_plant` command of the reference
(console script -> deepsignal_plant_amd.deepsignal_plant:main; the reference installs the same command name from
deepsignal_plant/deepsignal_plant.py).  The native library is built in-tree by `python -c 'import __graft_entry__ as g;
g.build()'` (hipcc --offload-arch=gfx950) and shipped as package data; nothing is compiled by setuptools."""
import os

from setuptools import setup

here = os.path.dirname(os.path.abspath(__file__))
version = {}
with open(os.path.join(here, "deepsignal_plant_amd", "_version.py")) as f:
    exec(f.read(), version)

setup(
    name="deepsignal-plant-amd",
    version=version["VERSION"],
    description="deepsignal_plant call_mods / call_freq / extract on AMD MI355X (hand-written HIP kernels behind a C ABI)",
    packages=["deepsignal_plant_amd", "deepsignal_plant_amd.utils"],
    package_data={"deepsignal_plant_amd": ["libdsp_amd.so"]},
    python_requires=">=3.8",
    install_requires=["numpy", "torch"],
    entry_points={"console_scripts": ["deepsignal_plant=deepsignal_plant_amd.deepsignal_plant:main"]},
)
