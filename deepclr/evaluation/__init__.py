from deepclr_amd.evaluation import Evaluator, MetricsContainer, Sequence, load_scenario

__all__ = ['Evaluator', 'MetricsContainer', 'Sequence', 'load_scenario']
