"""Import names of the reference package, served by deepclr_amd (MI355X kernels).

Exists so that the reference's entry scripts find what they import (`scripts/inference.py:9-13`,
`scripts/timing.py:6-10`): `deepclr.models`, `deepclr.config`, `deepclr.evaluation`, `deepclr.utils.logging`,
`deepclr.utils.tensor`, `deepclr.data.LabelType`. The dataset readers (`deepclr.data.create_input_dataflow`,
`make_data_loader`: LMDB + dataflow pipelines) and the training configuration tree (`load_config`) are outside the
forward hot path and raise with that explanation.
"""
