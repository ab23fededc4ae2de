from .karrasmodule import KarrasModule, KarrasModuleConfig  # noqa: F401
from .schedulers import Scheduler, EDMScheduler, VPScheduler, VEScheduler  # noqa: F401
from .integrators import (Integrator, EulerIntegrator, HeunIntegrator,  # noqa: F401
                          EulerMaruyamaIntegrator, KarrasIntegrator, name_to_integrator)
from .preconditioners import (KarrasPreconditioner, EDMPreconditioner,  # noqa: F401
                              NullPreconditioner, SR3Preconditioner, VPPreconditioner, VEPreconditioner)
from .noisesamplers import (NoiseSampler, EDMNoiseSampler, VPNoiseSampler, VENoiseSampler,  # noqa: F401
                            UniformNoiseSampler)
from .autoregressivesample import LatentSpaceAutoregressive  # noqa: F401
from .schedulingfunctions import (SchedulingFunctions, EDMSchedulingFunctions,  # noqa: F401
                                  VPSchedulingFunctions, VESchedulingFunctions,
                                  name_to_scheduling_functions)
from .flowfield import SIModule, SIModuleConfig, SIScheduler  # noqa: F401
