"""Build the parts of the REFERENCE that compile from their own few source files, into oracle/_ref/.

TEST INFRASTRUCTURE ONLY (see oracle/graspbal_oracle.c).  Only the KNN CPU path qualifies:
``KNN/Pytorch_CUDA_KNN/vision.cpp`` + ``cpu/knn_cpu.cpp`` (plain C++ over ATen, no CUDA).  The sources
are compiled where they lie under /root/reference — nothing is copied — and the outputs go to
``oracle/_ref/`` (git-ignored, but shipped to the GPU box by gpurun like other built .so files).

The CUDA extensions (PointNet/_ext_src, pointnet2_batch/src, KNN cuda/knn.cu) are UNBUILDABLE here:
they include cuda.h / ATen/cuda/CUDAContext.h and need nvcc, which this image lacks.
"""
import glob
import importlib.util
import os

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF = os.environ.get("GB_REFERENCE", "/root/reference")


def build_knn_ref(verbose=False):
    src_dir = os.path.join(REF, "KNN", "Pytorch_CUDA_KNN")
    if not os.path.isdir(src_dir):
        return None
    from torch.utils import cpp_extension
    os.makedirs(OUT, exist_ok=True)
    cpp_extension.load(
        name="knn_ref",
        sources=[os.path.join(src_dir, "vision.cpp"), os.path.join(src_dir, "cpu", "knn_cpu.cpp")],
        extra_include_paths=[src_dir],
        extra_cflags=["-O2", "-ffp-contract=off", "-w"],
        build_directory=OUT, with_cuda=False, verbose=verbose, is_python_module=True)
    return OUT


def load_knn_ref():
    """Import oracle/_ref/knn_ref.so (building it first when the reference tree is present)."""
    sos = glob.glob(os.path.join(OUT, "knn_ref*.so"))
    if not sos:
        if build_knn_ref() is None:
            raise FileNotFoundError("oracle/_ref/knn_ref.so missing and no reference tree to build it from")
        sos = glob.glob(os.path.join(OUT, "knn_ref*.so"))
    import torch  # noqa: F401  (the module links against libtorch)
    spec = importlib.util.spec_from_file_location("knn_ref", sos[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build_knn_ref(verbose=True))
