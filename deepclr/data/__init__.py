from deepclr_amd.labels import LabelType
from deepclr_amd.evaluation import DatasetType
from deepclr_amd.data import make_data_loader


def _no_reader(*_args, **_kwargs):
    raise RuntimeError("the reference's dataset readers (LMDB through dataflow) are outside the MI355X forward hot "
                       "path; feed clouds as tensors (deepclr_amd.preprocess prepares raw scans on the device), or "
                       "use make_data_loader with dataset_type: synthetic_kitti / synthetic_modelnet")


create_input_dataflow = make_dataflow = build_dataset = _no_reader

__all__ = ['LabelType', 'DatasetType', 'create_input_dataflow', 'make_data_loader', 'make_dataflow', 'build_dataset']
