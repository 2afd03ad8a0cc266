from deepclr_amd.evaluation import Evaluator, Sequence, load_scenario

__all__ = ['Evaluator', 'Sequence', 'load_scenario']
