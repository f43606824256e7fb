"""Drop-in for the reference's `dataset` module name (diffusion_training.py:13 `import dataset`): the MRI slice loader."""
from anoddpm_amd.dataset import MRIDataset, cycle, init_dataset_loader  # noqa: F401


def init_datasets(ROOT_DIR, args, device=None):
    """dataset.py:351-358.  `device` (an extension; default = the calling process's current HIP device) is where the volumes
    live and the loader kernels run -- one process per GPU passes its own."""
    return (MRIDataset(ROOT_DIR=f'{ROOT_DIR}DATASETS/Train/', img_size=args['img_size'], random_slice=args['random_slice'], device=device),
            MRIDataset(ROOT_DIR=f'{ROOT_DIR}DATASETS/Test/', img_size=args['img_size'], random_slice=args['random_slice'], device=device))
