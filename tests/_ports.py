"""A rendezvous port that is free NOW (bind to 0, read it back, close): fixed pid-derived ports collide with the ephemeral ports gloo's
own pair connections leave in TIME_WAIT between tests."""
import socket


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]
