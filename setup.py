"""Packaging of the MI355X N-HANS hot path: console scripts with the reference's names
(reference setup.py:44-50).  The package directory is `n-hans_amd/`, importable as `nhans_amd`."""
from setuptools import setup

setup(
    name="nhans-amd",
    version="0.1.0",
    description="N-HANS per-frame inference hot path on MI355X (gfx950)",
    packages=["nhans_amd"],
    package_dir={"nhans_amd": "n-hans_amd"},
    package_data={"nhans_amd": ["csrc/libnhans_hip.so"]},
    install_requires=["numpy", "scipy", "torch"],
    entry_points={"console_scripts": [
        "nhans_denoiser = nhans_amd.apply:main",
        "nhans_separator = nhans_amd.apply:main_separator",
        "load_denoiser = nhans_amd.load_model:main",
        "load_separator = nhans_amd.load_model:main_separator",
    ]},
)
