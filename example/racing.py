"""Racing (counterpart of the reference's example/racing.py main loop, rendering removed)."""
import _common  # noqa: F401
from envs.racing_controller import racing_controller
from envs.racing_env import RacingEnv


def main(max_steps: int = 500):
    env = RacingEnv()
    controller = racing_controller(env, debug=False)
    controller.set_cost_map(env._obstacle_map, env._lane_map)
    state = env.reset()
    for i in range(max_steps):
        action_seq, state_seq = controller.update(state, env.racing_center_path)
        state, is_goal_reached = env.step(action_seq[0, :])
        is_collisions = env.collision_check(state=state_seq)
        top_samples, top_weights = controller.get_top_samples(num_samples=300)
        if is_goal_reached:
            print("Goal Reached!")
            break
    print(f"{i + 1} steps, path index {controller.current_path_index}, state {state.tolist()}")


if __name__ == "__main__":
    main()
