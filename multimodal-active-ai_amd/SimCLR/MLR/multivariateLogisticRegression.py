"""Reference-compatible SimCLR/MLR/multivariateLogisticRegression.py (:6-13).
One nn.Linear used only by the linear-probe driver — outside the hot path
(SURVEY §2.1 marks it out of scope), kept as plain torch so that
Representation_Evaluation.py imports resolve."""
import torch.nn as nn


class LogisticRegression(nn.Module):
    def __init__(self, input_dim, output_dim):
        super().__init__()
        self.linear = nn.Linear(input_dim, output_dim)

    def forward(self, x):
        return self.linear(x)
