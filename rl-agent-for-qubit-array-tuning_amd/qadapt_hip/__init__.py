"""qadapt_hip -- MI355X-native batched quantum-dot tuning environment.

Host-side mirror of the reference's env surface (QuantumDeviceEnv /
MultiAgentEnvWrapper, src/qadapt/environment/) over a C-ABI HIP library
(csrc/ -> libqdsim.so).  There is no CPU fallback: importing the compute path
without the built library raises.
"""
__all__ = ["layout", "device_model"]
