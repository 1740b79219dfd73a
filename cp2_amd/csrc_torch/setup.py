"""In-tree build of the C++ autograd nodes:  python cp2_amd/csrc_torch/setup.py build_ext --inplace
(host code only -- plain g++ against the ROCm torch headers; the .so lands next to cp2_amd/__init__.py)."""
import os

from setuptools import setup
from torch.utils.cpp_extension import BuildExtension, CppExtension

here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
os.chdir(root)
rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
setup(
    name="cp2_amd_autograd_ext",
    ext_modules=[CppExtension(
        "cp2_amd._autograd_ext", [os.path.relpath(os.path.join(here, "autograd_ext.cpp"), root)],
        include_dirs=[os.path.join(rocm, "include")],
        define_macros=[("__HIP_PLATFORM_AMD__", "1"), ("USE_ROCM", "1")],
        extra_compile_args=["-O2", "-g0", "-std=c++17", "-Wno-deprecated-declarations"],
        libraries=["c10_hip", "torch_hip"],
        extra_link_args=["-Wl,-rpath," + os.path.join(os.path.dirname(__import__("torch").__file__), "lib")],
    )],
    cmdclass={"build_ext": BuildExtension.with_options(use_ninja=False)},
)
