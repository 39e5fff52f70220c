from .vae import VAESolver
from .intro import IntroSolver
