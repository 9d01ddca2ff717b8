"""DirectedGVAE sampler + KL (reference: DG_VAE/deepgate/digvae_model.py:105-190, trainer.py:145-148).

The reference's DG_VAE training path does not run (its `forward` calls the decoder without
edge_index and `Trainer.run_batch` expects `(hs, hf)`); what it defines — the four
mu/logstd heads, the reparameterised sample and the KL expression — is implemented here on the
HIP kernels, with `forward` completed in the obvious way (decode the sampled embeddings on the
graph's own edges)."""
import torch

from . import ops
from .digae_layer import DirectedInnerProductDecoder
from .sampling import negative_sampling

EPS = 1e-15
MAX_LOGSTD = 10


class DirectedGVAE(torch.nn.Module):
    def __init__(self, encoder, dim_hidden, decoder=None):
        super().__init__()
        self.encoder = encoder
        self.decoder = DirectedInnerProductDecoder() if decoder is None else decoder
        self.dim_hidden = dim_hidden
        self.fc_s_mu = torch.nn.Linear(dim_hidden, dim_hidden)
        self.fc_s_logstd = torch.nn.Linear(dim_hidden, dim_hidden)
        self.fc_t_mu = torch.nn.Linear(dim_hidden, dim_hidden)
        self.fc_t_logstd = torch.nn.Linear(dim_hidden, dim_hidden)
        self._klsum = None

    def sample(self, s, t, eps_s=None, eps_t=None, seed=None):
        """z = mu + exp(logstd) * eps for both embeddings (digvae_model.py:134-142).  eps_* inject the
        noise (parity tests); otherwise the kernel draws it from a counter-based generator."""
        self.s_mu, self.s_logstd = ops.linear(s, self.fc_s_mu.weight, self.fc_s_mu.bias), ops.linear(s, self.fc_s_logstd.weight, self.fc_s_logstd.bias)
        self.t_mu, self.t_logstd = ops.linear(t, self.fc_t_mu.weight, self.fc_t_mu.bias), ops.linear(t, self.fc_t_logstd.weight, self.fc_t_logstd.bias)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        sample_s, kls = ops.ReparamFn.apply(self.s_mu, self.s_logstd, eps_s, seed)
        sample_t, klt = ops.ReparamFn.apply(self.t_mu, self.t_logstd, eps_t, seed + 1)
        self._klsum = (kls, klt)
        return sample_s, sample_t

    def kl_loss(self):
        """(s_kl, t_kl) exactly as trainer.py:146-147 writes them: -0.5/N * mean_i sum_d(...)."""
        n = self.s_mu.shape[0]
        c = -0.5 / n / n
        return c * self._klsum[0], c * self._klsum[1]

    def encode(self, *args, **kwargs):
        return self.encoder(*args, **kwargs)

    def decode(self, *args, **kwargs):
        return self.decoder(*args, **kwargs)

    def forward(self, data):
        s, t = self.encoder(data.x, data.x, data.edge_index)
        sample_s, sample_t = self.sample(s, t)
        return self.decoder(sample_s, sample_t, data.edge_index)

    def recon_loss(self, s, t, pos_edge_index, neg_edge_index=None):
        s, t = self.sample(s, t)
        if neg_edge_index is None:
            neg_edge_index = negative_sampling(pos_edge_index, s.shape[0])
        st = torch.cat([s, t], dim=1)
        loss, counts, pred_bin = ops.ReconLossFn.apply(st, pos_edge_index, neg_edge_index, True)
        Ep, En = pos_edge_index.shape[1], neg_edge_index.shape[1]
        gt_bin = torch.zeros(Ep + En, dtype=torch.int32, device=s.device)
        gt_bin[:Ep] = 1
        return loss, pred_bin, gt_bin
