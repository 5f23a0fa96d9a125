class AutoencoderKL:  # type placeholder only: the oracle works in latent space
    pass
