"""
varnet_amd -- MI355X-native engine for VarNet's variational-loss training loop behind the
reference's VarNet / ADPDE / Domain constructor API.
"""
__version__ = '0.1.0'
