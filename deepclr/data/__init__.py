from deepclr_amd.labels import LabelType
from deepclr_amd.evaluation import DatasetType
from deepclr_amd.data import create_input_dataflow, make_data_loader


def _no_reader(*_args, **_kwargs):
    raise RuntimeError("the reference's training-side dataset builders (LMDB through dataflow, augmentation pipeline) are "
                       "outside the MI355X forward hot path; create_input_dataflow reads .npz sequence / pair files, "
                       "make_data_loader serves dataset_type: synthetic_kitti / synthetic_modelnet")


make_dataflow = build_dataset = _no_reader

__all__ = ['LabelType', 'DatasetType', 'create_input_dataflow', 'make_data_loader', 'make_dataflow', 'build_dataset']
