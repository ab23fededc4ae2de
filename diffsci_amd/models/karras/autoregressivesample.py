"""Autoregressive forecasting on top of ``KarrasModule.sample`` -- the ``LatentSpaceAutoregressive`` mixin of the
reference (diffsci/models/karras/autoregressivesample.py:18-292; mixed into KarrasModule at karrasmodule.py:403-407).

Host orchestration only: every forecast step is one ``self.sample(..., is_latent_shape=True,
return_in_latent_space=True)`` -- the captured HIP loop -- conditioned on ``y['y']``, the last ``cond_time`` frames
stacked along channels.  The window is kept exactly as the reference assembles it:

* each frame of the window is the prediction of sample 0 of the batch (``predictions[..][:, 0]``, :143-145,159);
* while fewer than ``cond_time`` predictions exist, the head of the window is cut from the window written on the
  PREVIOUS step (the loop re-reads ``y['y']`` after overwriting it, :147-161), not from the caller's frames.
"""
from typing import Dict, List, Optional

import torch


def window_frames(current, predictions, cond_time):
    """The conditioning frames [cond_time, C, h, w] for the next forecast.
    current: the window in force (frames [cond_time, C, h, w]); predictions: sample-0 frames so far, oldest first."""
    n = len(predictions)
    if n >= cond_time:
        return torch.stack(predictions[n - cond_time:], dim=0)
    need = cond_time - n                                    # frames still taken from the window in force ...
    return torch.cat([current[cond_time - need:], torch.stack(predictions, dim=0)], dim=0)   # ... its LAST `need`


class LatentSpaceAutoregressive:
    def autoregressive_sample(self, nsamples: int, latent_shape: List[int], nsteps_forecast: int, cond_time: int,
                              nsteps_diffusion: int = 50, y: Optional[Dict[str, torch.Tensor]] = None,
                              y_already_encoded: bool = False, guidance: float = 1.0,
                              maximum_batch_size: Optional[int] = None, return_intermediate: bool = False,
                              return_in_latent: bool = False) -> Dict[str, torch.Tensor]:
        """autoregressivesample.py:27-203.  Returns {'forecasts': [nsteps_forecast, nsamples, ...], 'final_forecast'
        (or 'final_forecast_latent' when return_in_latent)[, 'intermediate_latent']}."""
        with torch.inference_mode():
            if maximum_batch_size is not None:
                return self._autoregressive_sample_batched(nsamples, latent_shape, nsteps_forecast, cond_time,
                                                           nsteps_diffusion, y, y_already_encoded, guidance,
                                                           maximum_batch_size, return_intermediate, return_in_latent)
            y = dict(y) if y is not None else {}
            if "y" not in y:
                raise ValueError("y['y'] must be provided")
            if not y_already_encoded:
                y = self._encode_y_once(y)
            C, h, w = latent_shape
            frames = y["y"].reshape(cond_time, C, h, w).to(self.device)
            firsts, forecasts = [], []                      # sample 0 of every prediction; all predictions
            for step in range(nsteps_forecast):
                if step > 0:
                    frames = window_frames(frames, firsts, cond_time)
                    y["y"] = frames.reshape(cond_time * C, h, w)
                pred = self.sample(nsamples=nsamples, shape=latent_shape, y=y, guidance=guidance, nsteps=nsteps_diffusion,
                                   record_history=False, is_latent_shape=True, return_in_latent_space=True)
                forecasts.append(pred)
                firsts.append(pred[0])
            latent = torch.stack(forecasts, dim=0)
            if return_in_latent:
                return {"forecasts": latent, "final_forecast_latent": latent[-1]}
            flat = self.decode(latent.reshape(nsteps_forecast * nsamples, C, h, w), y, record_history=False)
            if isinstance(flat, tuple):
                flat = flat[0]
            pixel = flat.view(nsteps_forecast, nsamples, *flat.shape[1:])
            result = {"forecasts": pixel, "final_forecast": pixel[-1]}
            if return_intermediate:
                result["intermediate_latent"] = latent
            return result

    def _encode_y_once(self, y):
        """autoregressivesample.py:205-230: only modules that encode their condition (encode_y) touch y, through one
        encode() of a dummy batch; the shape of that batch is the reference's (1 x 3 x 128 x 128)."""
        if not getattr(self, "encode_y", False) or "y" not in y:
            return y
        dummy = torch.zeros(1, 3, 128, 128, device=self.device)
        try:
            out = self.encode(dummy, y, record_history=False)
        except Exception as e:  # the reference falls back to the raw condition, with a message
            print(f"Warning: encoding y failed with {e}, using original y")
            return y
        if not isinstance(out, tuple):
            return y
        res = dict(y)
        res.update(out[1])
        if res["y"].shape[0] == 1:
            res["y"] = res["y"].squeeze(0)
        return res

    def _autoregressive_sample_batched(self, nsamples, latent_shape, nsteps_forecast, cond_time, nsteps_diffusion, y,
                                       y_already_encoded, guidance, maximum_batch_size, return_intermediate,
                                       return_in_latent):
        """autoregressivesample.py:232-284: independent runs over near-equal minibatches, concatenated over samples."""
        parts = [self.autoregressive_sample(b, latent_shape, nsteps_forecast, cond_time, nsteps_diffusion, y,
                                            y_already_encoded, guidance, maximum_batch_size=None,
                                            return_intermediate=return_intermediate, return_in_latent=return_in_latent)
                 for b in self._get_minibatch_sizes(nsamples, maximum_batch_size)]
        result = {"forecasts": torch.cat([p["forecasts"] for p in parts], dim=1)}
        if "final_forecast" in parts[0]:
            result["final_forecast"] = torch.cat([p["final_forecast"] for p in parts], dim=0)
        if return_intermediate and "intermediate_latent" in parts[0]:
            result["intermediate_latent"] = torch.cat([p["intermediate_latent"] for p in parts], dim=1)
        return result

    def _get_minibatch_sizes(self, total: int, max_size: int) -> List[int]:
        """autoregressivesample.py:286-291: ceil(total / max_size) batches whose sizes differ by at most one."""
        n = (total + max_size - 1) // max_size
        base, extra = divmod(total, n)
        return [base + (1 if i < extra else 0) for i in range(n)]
