from .karrasmodule import KarrasModule, KarrasModuleConfig  # noqa: F401
from .schedulers import Scheduler, EDMScheduler  # noqa: F401
from .integrators import (Integrator, EulerIntegrator, HeunIntegrator,  # noqa: F401
                          EulerMaruyamaIntegrator, KarrasIntegrator, name_to_integrator)
from .preconditioners import (KarrasPreconditioner, EDMPreconditioner,  # noqa: F401
                              NullPreconditioner, SR3Preconditioner)
from .noisesamplers import NoiseSampler, EDMNoiseSampler  # noqa: F401
from .schedulingfunctions import (SchedulingFunctions, EDMSchedulingFunctions,  # noqa: F401
                                  name_to_scheduling_functions)
