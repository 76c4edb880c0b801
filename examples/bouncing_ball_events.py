"""Terminal event detection (the reference's examples/bouncing_ball.rs): a ball with quadratic drag, stopped at impact."""
from ivp_amd import BouncingBall, Options, solve_ivp

sol = solve_ivp(BouncingBall(gravity=9.81, drag=0.02), 0.0, 10.0, [10.0, 5.0], Options(method="DOPRI5", rtol=1e-8, atol=1e-10))
print(f"status {sol.status.name}: impact at t = {sol.t_events[0][0]:.6f} s with v = {sol.y_events[0][0][1]:.4f} m/s after {sol.naccpt} steps")
assert sol.status.name == "UserInterrupt" and abs(sol.y_events[0][0][0]) < 1e-9
