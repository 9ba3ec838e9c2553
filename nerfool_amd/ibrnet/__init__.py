"""Host-side mirror of the reference's `ibrnet` package for the attack path: same callables, argument meaning and
return schemas, every tensor operation dispatched to libnerfool_hip.so."""
